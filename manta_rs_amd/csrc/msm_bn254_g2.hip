// MSM engine instantiation: Bn254, G2.
#include "msm_impl.h"
namespace mg {
GroupEngine *make_engine_bn254_g2() { return new GroupEngineT<Bn254, 0, 2>(); }
} // namespace mg
