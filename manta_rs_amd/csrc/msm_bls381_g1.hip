// MSM engine instantiation: Bls381, G1.
#include "msm_impl.h"
namespace mg {
GroupEngine *make_engine_bls381_g1() { return new GroupEngineT<Bls381, 1, 1>(); }
} // namespace mg
