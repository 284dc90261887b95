#include "linalg.hpp"
namespace orc {
int gs_arm_dev(const MatView &, const double *, double *, uint64_t, double, int, Arena &) { return ORC_ERR_UNSUPPORTED_SOLVER; }
}
