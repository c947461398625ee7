// Fr engine instantiation (Bn254FrCfg).
#include "fr_impl.h"
namespace mg {
FrEngine *make_fr_engine_bn254() { return new FrEngineT<Bn254FrCfg>(); }
} // namespace mg
