// MSM engine instantiation: Bls381, G2.
#include "msm_impl.h"
namespace mg {
GroupEngine *make_engine_bls381_g2() { return new GroupEngineT<Bls381, 1, 2>(); }
} // namespace mg
