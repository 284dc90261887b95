"""settings::* of the reference (src/lib.rs:8-202) as plain values for the C ABI."""
import ctypes as C

from ._lib import lib


class NumericalSettings(C.Structure):
    """NumericalSettings + MatrixSolverSettings (lib.rs:14-56) flattened = OrcSettings."""
    _fields_ = [
        ("momentum", C.c_int32), ("diffusion", C.c_int32), ("pressure_interpolation", C.c_int32),
        ("velocity_interpolation", C.c_int32), ("gradient_reconstruction", C.c_int32),
        ("solver_type", C.c_int32), ("preconditioner", C.c_int32), ("q1_compat", C.c_int32),
        ("iterations", C.c_uint64), ("momentum_relaxation", C.c_double), ("pressure_relaxation", C.c_double),
        ("relaxation", C.c_double), ("relative_convergence_threshold", C.c_double),
        ("frozen_diagonals", C.c_int32), ("breakdown_guard", C.c_int32),
        ("reduction_order", C.c_int32), ("reserved0", C.c_int32),
    ]

    @classmethod
    def default(cls, **overrides):
        """NumericalSettings::default() (lib.rs:58-86), struct-update style overrides."""
        s = cls()
        lib().orc_settings_default(C.byref(s))
        for k, v in overrides.items():
            if not hasattr(s, k):
                raise AttributeError(k)
            setattr(s, k, v)
        return s


class MomentumDiscretization:  # lib.rs:95-118
    UD, CD1, CD2, TVD_LUD, TVD_QUICK, TVD_UMIST, TVD_UD, TVD_CD1 = range(8)


class PressureInterpolation:  # lib.rs:125-133
    Linear, LinearWeighted, Standard, SecondOrder, NoInterpolation = range(5)


class VelocityInterpolation:  # lib.rs:135-146
    Linear, LinearWeighted, RhieChow, NoInterpolation = range(4)


class SolutionMethod:  # lib.rs:171-179 (+ new-build extensions)
    GaussSeidel, Jacobi, Multigrid, BiCGSTAB = range(4)
    MulticolorGS, BiCGSTAB_GS, Multigrid_GS = 16, 17, 18


class ReductionOrder:  # include/orc_types.h OrcReductionOrder
    Tree, Reference = 0, 1


class PreconditionMethod:  # lib.rs:181-185
    NoPreconditioner, Jacobi = 0, 1


class FaceConditionTypes:  # mesh.rs:25-65
    Interior, Wall, PressureInlet, PressureOutlet, Symmetry, VelocityInlet = 2, 3, 4, 5, 7, 10
