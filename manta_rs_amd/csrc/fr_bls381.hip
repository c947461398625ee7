// Fr engine instantiation (Bls381FrCfg).
#include "fr_impl.h"
namespace mg {
FrEngine *make_fr_engine_bls381() { return new FrEngineT<Bls381FrCfg>(); }
} // namespace mg
