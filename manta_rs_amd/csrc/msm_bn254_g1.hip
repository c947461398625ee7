// MSM engine instantiation: Bn254, G1.
#include "msm_impl.h"
namespace mg {
GroupEngine *make_engine_bn254_g1() { return new GroupEngineT<Bn254, 0, 1>(); }
} // namespace mg
