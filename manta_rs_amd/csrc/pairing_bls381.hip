#include "pairing_impl.h"
namespace mg {
PairingEngine *make_pairing_engine_bls381() { return new PairingEngineT<Bls381Pairing>(); }
} // namespace mg
