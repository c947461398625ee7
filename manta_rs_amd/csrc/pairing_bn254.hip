#include "pairing_impl.h"
namespace mg {
PairingEngine *make_pairing_engine_bn254() { return new PairingEngineT<Bn254Pairing>(); }
} // namespace mg
