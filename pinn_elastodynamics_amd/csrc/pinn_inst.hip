// Instantiation unit: compiled once per line of pinn_variants.def with
//   -DPINN_INST_OP=<F16|BF16> -DPINN_INST_SPLIT=<1|3> -DPINN_INST_WIDTH=<32|64|96|128|160>
#include "pinn_host.hpp"

#define PINN_CAT_(a, b, c, d) a##b##_##c##_##d
#define PINN_CAT(a, b, c, d) PINN_CAT_(a, b, c, d)
#define PINN_OPT_(o) Op##o
#define PINN_OPT(o) PINN_OPT_(o)

namespace pinn {
const Impl* PINN_CAT(impl_, PINN_INST_OP, PINN_INST_SPLIT, PINN_INST_WIDTH)() {
    return Host<PINN_OPT(PINN_INST_OP), PINN_INST_SPLIT, PINN_INST_WIDTH>::impl();
}
}  // namespace pinn
