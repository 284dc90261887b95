/* orc_types.h — plain-C value types shared by the C-ABI (include/orc_amd.h) and the
 * test oracle (oracle/).  Values only, no code.
 *
 * Every enum mirrors one Rust enum of the reference (ORC v0.3.0); the numeric values are
 * ours (Rust enums without #[repr] have no ABI), the names and meaning are the reference's.
 * Citations are file:line into /root/reference.
 */
#ifndef ORC_TYPES_H
#define ORC_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* settings::MomentumDiscretization (src/lib.rs:95-118).  The reference's `TVD(fn(f64)->f64)`
 * carries a bare function pointer; across the C ABI it becomes a limiter tag. */
enum OrcMomentumDiscretization {
    ORC_MOMENTUM_UD = 0,        /* lib.rs:98  */
    ORC_MOMENTUM_CD1 = 1,       /* lib.rs:100 */
    ORC_MOMENTUM_CD2 = 2,       /* lib.rs:102 — reference panics "unsupported momentum scheme" (discretization.rs:287) */
    ORC_MOMENTUM_TVD_LUD = 3,   /* lib.rs:109  psi(r) = r */
    ORC_MOMENTUM_TVD_QUICK = 4, /* lib.rs:110  psi(r) = (3 + r) / 4 */
    ORC_MOMENTUM_TVD_UMIST = 5, /* lib.rs:111-118 */
    ORC_MOMENTUM_TVD_UD = 6,    /* lib.rs:107  psi = 0 (private const in the reference) */
    ORC_MOMENTUM_TVD_CD1 = 7    /* lib.rs:108  psi = 1 (private const in the reference) */
};

/* settings::DiffusionScheme (lib.rs:120-123) */
enum OrcDiffusionScheme { ORC_DIFFUSION_CD = 0 };

/* settings::PressureInterpolation (lib.rs:125-133) */
enum OrcPressureInterpolation {
    ORC_PINTERP_LINEAR = 0,
    ORC_PINTERP_LINEAR_WEIGHTED = 1,
    ORC_PINTERP_STANDARD = 2, /* reference panics (solver.rs:1136) */
    ORC_PINTERP_SECOND_ORDER = 3,
    ORC_PINTERP_NONE = 4
};

/* settings::VelocityInterpolation (lib.rs:135-146) */
enum OrcVelocityInterpolation {
    ORC_VINTERP_LINEAR = 0,
    ORC_VINTERP_LINEAR_WEIGHTED = 1,
    ORC_VINTERP_RHIE_CHOW = 2,
    ORC_VINTERP_NONE = 3
};

/* settings::GradientReconstructionMethods (lib.rs:154-162) */
enum OrcGradientReconstruction {
    ORC_GRAD_GREEN_GAUSS_CELL = 0,
    ORC_GRAD_GREEN_GAUSS_NODE = 1, /* reference panics (solver.rs:901) */
    ORC_GRAD_LEAST_SQUARES = 2,    /* solver.rs:803-869, 903-947: normal equations per cell, 3x3 inverse */
    ORC_GRAD_NONE = 3
};

/* settings::SolutionMethod (lib.rs:171-179) */
enum OrcSolutionMethod {
    ORC_SOLVER_GAUSS_SEIDEL = 0, /* reference: dense row scan then panic!("Gauss-Seidel out for maintenance") (linear_algebra.rs:219-246) */
    ORC_SOLVER_JACOBI = 1,
    ORC_SOLVER_MULTIGRID = 2,
    ORC_SOLVER_BICGSTAB = 3,
    /* --- new-build extensions, no reference counterpart --- */
    ORC_SOLVER_MULTICOLOR_GS = 16,         /* multicolour Gauss-Seidel sweeps */
    ORC_SOLVER_BICGSTAB_GS_PRECOND = 17,   /* right-preconditioned BiCGSTAB, M = one multicolour GS sweep */
    ORC_SOLVER_MULTIGRID_GS = 18           /* Multigrid arm with multicolour GS as the smoother */
};

/* settings::PreconditionMethod (lib.rs:181-185) */
enum OrcPreconditionMethod { ORC_PRECOND_NONE = 0, ORC_PRECOND_JACOBI = 1 };

/* mesh::FaceConditionTypes (mesh.rs:25-42); values are the TGRID codes of mesh.rs:51-65 */
enum OrcFaceConditionType {
    ORC_BC_INTERIOR = 2,
    ORC_BC_WALL = 3,
    ORC_BC_PRESSURE_INLET = 4,
    ORC_BC_PRESSURE_OUTLET = 5,
    ORC_BC_SYMMETRY = 7,
    ORC_BC_PERIODIC_SHADOW = 8,
    ORC_BC_PRESSURE_FAR_FIELD = 9,
    ORC_BC_VELOCITY_INLET = 10,
    ORC_BC_PERIODIC = 12,
    ORC_BC_POROUS_JUMP = 14,
    ORC_BC_MASS_FLOW_INLET = 20,
    ORC_BC_INTERFACE = 24,
    ORC_BC_PARENT = 31,
    ORC_BC_OUTFLOW = 36,
    ORC_BC_AXIS = 37
};

/* Status codes: one per panic! site on the path (SURVEY §5 "Failure detection").
 * A Rust shim turns non-zero back into panic!(orc_status_string(code)). */
enum OrcStatus {
    ORC_OK = 0,
    ORC_ERR_SOLUTION_DIVERGED = 1,     /* solver.rs:217-221 "solution diverged" */
    ORC_ERR_MULTIGRID_DIVERGED = 2,    /* linear_algebra.rs:103-105 */
    ORC_ERR_JACOBI_NAN = 3,            /* linear_algebra.rs:192-196 "diverged" */
    ORC_ERR_JACOBI_TOO_LARGE = 4,      /* linear_algebra.rs:214-216 */
    ORC_ERR_GS_MAINTENANCE = 5,        /* linear_algebra.rs:245 */
    ORC_ERR_STRUCTURAL_ZERO = 6,       /* lib.rs:664-666 */
    ORC_ERR_UNSUPPORTED_BC = 7,        /* discretization.rs:114-117, solver.rs:1001,1100,1148,1209-1212 */
    ORC_ERR_UNSUPPORTED_SCHEME = 8,    /* discretization.rs:50,287; solver.rs:224,870,901,948,994,1097,1136,1145 */
    ORC_ERR_UNSUPPORTED_SOLVER = 9,    /* linear_algebra.rs:297 */
    ORC_ERR_BAD_ARGUMENT = 10,
    ORC_ERR_NO_DEVICE = 11,            /* HIP runtime/device missing: the product never falls back to a CPU path */
    ORC_ERR_HIP = 12,
    ORC_ERR_IO = 13,
    ORC_ERR_COMM = 14,
    ORC_ERR_MESH_FORMAT = 15,          /* io.rs:32-515: any of read_mesh's expect()/panic! sites; orc_last_error() names file:line and the reference's message */
    ORC_ERR_ZONE_NOT_FOUND = 16,       /* mesh.rs:189-195 "face zone '{zone_name}' should exist in mesh" */
    ORC_ERR_NO_BOUNDARY_CONDITIONS = 17, /* solver.rs:770 "You must set boundary conditions." */
    ORC_ERR_SINGULAR_MATRIX = 18       /* solver.rs:850,943: `a.try_inverse().unwrap()` on a singular least-squares normal matrix */
};

/* Association of the solvers' dot products and norms (linear_algebra.rs:97,202,253,257,261,265).
 * TREE: per-workgroup wave-shuffle trees folded in a fixed order — deterministic, fast, the product default.
 * REFERENCE: nalgebra 0.32.4's `dotx` order (eight running accumulators over blocks of 8, then the tail), evaluated by
 * one wavefront — slow, but every iterate of BiCGSTAB / Jacobi / the Multigrid arm is then bit-identical to the
 * reference's arithmetic at any iteration count (verification mode; single GPU only). */
enum OrcReductionOrder { ORC_REDUCTION_TREE = 0, ORC_REDUCTION_REFERENCE = 1 };

/* Cell orders for orc_mesh_partition: ORC's own numbering (io.rs:404-438), reverse Cuthill-McKee over the face-neighbour
 * graph, or sorted along the longest extent of the domain. */
enum OrcCellOrdering { ORC_ORDER_ORC = 0, ORC_ORDER_RCM = 1, ORC_ORDER_GEOMETRIC = 2 };

/* settings::NumericalSettings + settings::MatrixSolverSettings (lib.rs:14-56), flattened.
 * orc_settings_default() fills in lib.rs:58-86. */
typedef struct OrcSettings {
    int32_t momentum;                /* OrcMomentumDiscretization; default CD1   (lib.rs:62) */
    int32_t diffusion;               /* OrcDiffusionScheme;        default CD    (lib.rs:63) */
    int32_t pressure_interpolation;  /* default SecondOrder (lib.rs:64) */
    int32_t velocity_interpolation;  /* default RhieChow    (lib.rs:65) */
    int32_t gradient_reconstruction; /* default GreenGauss(CellBased) (lib.rs:66-68) */
    int32_t solver_type;             /* default Multigrid   (lib.rs:79) */
    int32_t preconditioner;          /* default Jacobi      (lib.rs:83) */
    int32_t q1_compat;               /* 1 = reproduce `f64 * Vector` z:=y bug (lib.rs:540-548); default 1 */
    uint64_t iterations;             /* default 50   (lib.rs:80) */
    double momentum_relaxation;      /* default 0.5  (lib.rs:70) */
    double pressure_relaxation;      /* default 0.01 (lib.rs:69) */
    double relaxation;               /* default 0.5  (lib.rs:81) */
    double relative_convergence_threshold; /* default 1e-3 (lib.rs:82) */
    int32_t frozen_diagonals;        /* 0 = reference's in-place (order dependent) diagonal reads — oracle only;
                                        1 = all Rhie-Chow reads see last iteration's diagonals (what the device computes) */
    int32_t breakdown_guard;         /* new-build extension, default 1: BiCGSTAB stops updating x when a denominator of its
                                        recurrences (rho, r_hat.nu, t.t, omega) is exactly 0 or non-finite — the only cases
                                        in which the reference (no guard, linear_algebra.rs:255-268) yields NaN and panics
                                        "solution diverged".  0 = reference behaviour (NaN propagates).  Every solve in
                                        which the guard fired is counted: orc_breakdown_guard_events(). */
    int32_t reduction_order;         /* OrcReductionOrder; default TREE */
    int32_t reserved0;
} OrcSettings;

#ifdef __cplusplus
}
#endif
#endif /* ORC_TYPES_H */
