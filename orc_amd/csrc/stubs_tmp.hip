#include "linalg.hpp"
namespace orc {
int multigrid_arm_dev(const MatView &, const double *, double *, uint64_t, double, double, int, Arena &, SolveStats *, int) { return ORC_ERR_UNSUPPORTED_SOLVER; }
int gs_arm_dev(const MatView &, const double *, double *, uint64_t, double, int, Arena &) { return ORC_ERR_UNSUPPORTED_SOLVER; }
}
