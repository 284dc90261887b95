//! dump_golden.rs — reference-side golden dumper for the MI355X hot-path build (orc_amd).
//!
//! NOT part of the product: this file is meant to be dropped into a checkout of reidprichard/ORC (v0.3.0) as
//! `examples/dump_golden.rs` on a machine that has a Rust toolchain (the build container has none), where it calls ORC's
//! OWN functions on the fixtures and inputs this repository ships and writes every array the parity tests compare:
//!
//!     discretization::build_momentum_diffusion_matrix      (src/discretization.rs:39)
//!     discretization::initialize_momentum_matrix           (src/discretization.rs:450)
//!     discretization::build_momentum_advection_matrices    (src/discretization.rs:134)   two consecutive assemblies
//!     discretization::build_pressure_correction_matrices   (src/discretization.rs:359)
//!     linear_algebra::iterative_solve                      (src/linear_algebra.rs:144)   Jacobi / BiCGSTAB / Multigrid arms
//!     solver::solve_steady                                 (src/solver.rs:26)
//!     solver::initialize_flow                              (src/solver.rs:246)
//!
//! Recipe (see rust/README.md):
//!     python rust/export_inputs.py  rust/inputs            # in this repository: meshes, seeded fields, test matrices
//!     cp rust/dump_golden.rs <ORC>/examples/dump_golden.rs
//!     (cd <ORC> && cargo run --release --example dump_golden -- <repo>/rust/inputs <repo>/tests/golden/reference_dump)
//!     python rust/compare_with_oracle.py tests/golden/reference_dump     # bit-for-bit against oracle/
//!
//! File format: one file per array, raw little-endian — `*.f64` (IEEE-754 doubles), `*.i64`; `*.txt` for statuses
//! ("ok" or the panic message).  A panicking reference call (linear_algebra.rs:103-105 "Multigrid diverged",
//! solver.rs:217-221 "solution diverged", ...) is caught and recorded, as the C ABI turns the same sites into status codes.
use std::fs;
use std::io::Write;
use std::panic::{catch_unwind, AssertUnwindSafe};
use std::path::{Path, PathBuf};

use nalgebra::DVector;
use nalgebra_sparse::CsrMatrix;

use orc::discretization::{
    build_momentum_advection_matrices, build_momentum_diffusion_matrix, build_pressure_correction_matrices,
    initialize_momentum_matrix,
};
use orc::io::read_mesh;
use orc::linear_algebra::iterative_solve;
use orc::mesh::{FaceConditionTypes, Mesh};
use orc::numerical_types::{Float, Uint, Vector};
use orc::settings::*;
use orc::solver::{initialize_flow, solve_steady};

fn read_f64(path: &Path) -> Vec<f64> {
    let bytes = fs::read(path).unwrap_or_else(|e| panic!("cannot read {}: {e}", path.display()));
    bytes.chunks_exact(8).map(|c| f64::from_le_bytes(c.try_into().unwrap())).collect()
}

fn read_i64(path: &Path) -> Vec<i64> {
    let bytes = fs::read(path).unwrap_or_else(|e| panic!("cannot read {}: {e}", path.display()));
    bytes.chunks_exact(8).map(|c| i64::from_le_bytes(c.try_into().unwrap())).collect()
}

fn write_f64(path: &Path, data: &[f64]) {
    let mut f = fs::File::create(path).unwrap_or_else(|e| panic!("cannot write {}: {e}", path.display()));
    for x in data {
        f.write_all(&x.to_le_bytes()).unwrap();
    }
}

fn write_i64(path: &Path, data: &[i64]) {
    let mut f = fs::File::create(path).unwrap_or_else(|e| panic!("cannot write {}: {e}", path.display()));
    for x in data {
        f.write_all(&x.to_le_bytes()).unwrap();
    }
}

fn write_text(path: &Path, text: &str) {
    fs::write(path, text).unwrap_or_else(|e| panic!("cannot write {}: {e}", path.display()));
}

fn panic_text(e: Box<dyn std::any::Any + Send>) -> String {
    if let Some(s) = e.downcast_ref::<&str>() {
        s.to_string()
    } else if let Some(s) = e.downcast_ref::<String>() {
        s.clone()
    } else {
        "panic".to_string()
    }
}

fn dump_csr_pattern(dir: &Path, name: &str, a: &CsrMatrix<Float>) {
    let rp: Vec<i64> = a.row_offsets().iter().map(|&x| x as i64).collect();
    let ci: Vec<i64> = a.col_indices().iter().map(|&x| x as i64).collect();
    write_i64(&dir.join(format!("{name}_row_ptr.i64")), &rp);
    write_i64(&dir.join(format!("{name}_col.i64")), &ci);
}

// ---- boundary conditions of the parity cases (tests/helpers.py in the orc_amd repository; tests.rs:60-76, main.rs:282-293)
fn channel_bcs(mesh: &mut Mesh) {
    let has_top = mesh.face_zones.values().any(|z| z.name == "TOP_WALL");
    if has_top {
        mesh.get_face_zone("TOP_WALL").zone_type = FaceConditionTypes::Wall;
        mesh.get_face_zone("TOP_WALL").vector_value = Vector { x: 0., y: 0., z: 0. };
        mesh.get_face_zone("BOTTOM_WALL").zone_type = FaceConditionTypes::Wall;
    } else {
        mesh.get_face_zone("WALL").zone_type = FaceConditionTypes::Wall;
    }
    mesh.get_face_zone("INLET").zone_type = FaceConditionTypes::PressureInlet;
    mesh.get_face_zone("INLET").scalar_value = -5.0 * 0.002;
    mesh.get_face_zone("OUTLET").zone_type = FaceConditionTypes::PressureOutlet;
    mesh.get_face_zone("OUTLET").scalar_value = 0.;
    mesh.get_face_zone("PERIODIC_-Z").zone_type = FaceConditionTypes::Symmetry;
    mesh.get_face_zone("PERIODIC_+Z").zone_type = FaceConditionTypes::Symmetry;
}

fn cube_bcs(mesh: &mut Mesh) {
    mesh.get_face_zone("INLET").zone_type = FaceConditionTypes::PressureInlet;
    mesh.get_face_zone("INLET").scalar_value = 1.0;
    mesh.get_face_zone("OUTLET").zone_type = FaceConditionTypes::PressureOutlet;
    mesh.get_face_zone("OUTLET").scalar_value = 0.;
    mesh.get_face_zone("PERIODIC_-Z").zone_type = FaceConditionTypes::Wall;
    mesh.get_face_zone("PERIODIC_+Z").zone_type = FaceConditionTypes::Wall;
}

fn cube_bcs_mixed(mesh: &mut Mesh) {
    mesh.get_face_zone("INLET").zone_type = FaceConditionTypes::VelocityInlet;
    mesh.get_face_zone("INLET").scalar_value = 0.;
    mesh.get_face_zone("INLET").vector_value = Vector { x: 0.3, y: 0.02, z: -0.01 };
    mesh.get_face_zone("OUTLET").zone_type = FaceConditionTypes::PressureOutlet;
    mesh.get_face_zone("OUTLET").scalar_value = 0.25;
    mesh.get_face_zone("PERIODIC_-Z").zone_type = FaceConditionTypes::Symmetry;
    mesh.get_face_zone("PERIODIC_+Z").zone_type = FaceConditionTypes::Wall;
    mesh.get_face_zone("PERIODIC_+Z").vector_value = Vector { x: 0.1, y: 0.05, z: 0.0 };
}

fn settings_with(momentum: MomentumDiscretization, solver: SolutionMethod, iterations: Uint) -> NumericalSettings {
    NumericalSettings {
        momentum,
        matrix_solver: MatrixSolverSettings { solver_type: solver, iterations, ..MatrixSolverSettings::default() },
        ..NumericalSettings::default()
    }
}

fn load_fields(dir: &Path) -> (DVector<Float>, DVector<Float>, DVector<Float>, DVector<Float>) {
    (
        DVector::from_vec(read_f64(&dir.join("u0.f64"))),
        DVector::from_vec(read_f64(&dir.join("v0.f64"))),
        DVector::from_vec(read_f64(&dir.join("w0.f64"))),
        DVector::from_vec(read_f64(&dir.join("p0.f64"))),
    )
}

/// The assembly arrays of tests/golden/<case>.npz, "faithful" (= the reference's own in-place) keys.
fn dump_assembly(case: &str, mesh_file: &str, bcs: fn(&mut Mesh), momentum: MomentumDiscretization, inputs: &Path, out: &Path) {
    let dir = out.join(case);
    fs::create_dir_all(&dir).unwrap();
    let mut mesh = read_mesh(inputs.join("meshes").join(mesh_file).to_str().unwrap());
    bcs(&mut mesh);
    let (u, v, w, p) = load_fields(&inputs.join(case));
    let (rho, mu): (Float, Float) = (1000.0, 1e-3);
    let s = settings_with(momentum, SolutionMethod::Multigrid, 50);
    let (a_di, b_u_di, b_v_di, b_w_di) = build_momentum_diffusion_matrix(&mesh, s.diffusion, mu);
    dump_csr_pattern(&dir, "pattern", &a_di);
    write_f64(&dir.join("a_di.f64"), a_di.values());
    write_f64(&dir.join("b_u_di.f64"), b_u_di.as_slice());
    write_f64(&dir.join("b_v_di.f64"), b_v_di.as_slice());
    write_f64(&dir.join("b_w_di.f64"), b_w_di.as_slice());
    let mut a_u = initialize_momentum_matrix(&mesh);
    let mut a_v = initialize_momentum_matrix(&mesh);
    let mut a_w = initialize_momentum_matrix(&mesh);
    write_f64(&dir.join("a_init.f64"), a_u.values());
    let n = mesh.cells.len();
    let (mut b_u, mut b_v, mut b_w) = (DVector::zeros(n), DVector::zeros(n), DVector::zeros(n));
    // two consecutive assemblies: the first reads the diagonal 1.0 of initialize_momentum_matrix in Rhie-Chow, the second the
    // assembled diagonals, in place (discretization.rs:182-197, 340-351)
    for it in 1..=2 {
        let (pe_avg, pe_min, pe_max) = build_momentum_advection_matrices(
            &mut a_u, &mut a_v, &mut a_w, &mut b_u, &mut b_v, &mut b_w, &a_di, &mesh, &u, &v, &w, &p, s.momentum,
            s.velocity_interpolation, s.pressure_interpolation, s.gradient_reconstruction, rho,
        );
        write_f64(&dir.join(format!("a_u_it{it}.f64")), a_u.values());
        write_f64(&dir.join(format!("a_v_it{it}.f64")), a_v.values());
        write_f64(&dir.join(format!("a_w_it{it}.f64")), a_w.values());
        write_f64(&dir.join(format!("b_u_it{it}.f64")), b_u.as_slice());
        write_f64(&dir.join(format!("b_v_it{it}.f64")), b_v.as_slice());
        write_f64(&dir.join(format!("b_w_it{it}.f64")), b_w.as_slice());
        write_f64(&dir.join(format!("peclet_it{it}.f64")), &[pe_avg, pe_min, pe_max]);
    }
    let system = build_pressure_correction_matrices(&mesh, &u, &v, &w, &p, &a_u, &a_v, &a_w, &s, rho);
    dump_csr_pattern(&dir, "p_pattern", &system.a);
    write_f64(&dir.join("a_p.f64"), system.a.values());
    write_f64(&dir.join("b_p.f64"), system.b.as_slice());
    println!("assembly dumped: {case}");
}

/// solve_steady from the same start: fields after `iterations` SIMPLE iterations, or the panic text.
fn dump_solve_steady(case: &str, tag: &str, mesh_file: &str, bcs: fn(&mut Mesh), s: NumericalSettings, iterations: Uint, inputs: &Path, out: &Path) {
    let dir = out.join(case);
    fs::create_dir_all(&dir).unwrap();
    let mut mesh = read_mesh(inputs.join("meshes").join(mesh_file).to_str().unwrap());
    bcs(&mut mesh);
    let (mut u, mut v, mut w, mut p) = load_fields(&inputs.join(case));
    let result = catch_unwind(AssertUnwindSafe(|| {
        solve_steady(&mut mesh, &mut u, &mut v, &mut w, &mut p, &s, 1000.0, 1e-3, iterations, iterations.max(1));
    }));
    match result {
        Ok(()) => {
            write_text(&dir.join(format!("solve_steady_{tag}_status.txt")), "ok");
            write_f64(&dir.join(format!("solve_steady_{tag}_u.f64")), u.as_slice());
            write_f64(&dir.join(format!("solve_steady_{tag}_v.f64")), v.as_slice());
            write_f64(&dir.join(format!("solve_steady_{tag}_w.f64")), w.as_slice());
            write_f64(&dir.join(format!("solve_steady_{tag}_p.f64")), p.as_slice());
        }
        Err(e) => write_text(&dir.join(format!("solve_steady_{tag}_status.txt")), &panic_text(e)),
    }
    println!("solve_steady dumped: {case} {tag}");
}

/// iterative_solve on a CSR system exported by rust/export_inputs.py (row_ptr, col, values, b, x0).
fn dump_iterative_solve(name: &str, inputs: &Path, out: &Path) {
    let src = inputs.join("systems").join(name);
    let dir = out.join("systems").join(name);
    fs::create_dir_all(&dir).unwrap();
    let rp: Vec<usize> = read_i64(&src.join("row_ptr.i64")).iter().map(|&x| x as usize).collect();
    let ci: Vec<usize> = read_i64(&src.join("col.i64")).iter().map(|&x| x as usize).collect();
    let values = read_f64(&src.join("values.f64"));
    let b = DVector::from_vec(read_f64(&src.join("b.f64")));
    let x0 = read_f64(&src.join("x0.f64"));
    let n = b.len();
    let a = CsrMatrix::try_from_csr_data(n, n, rp, ci, values).expect("valid CSR data");
    let threshold: Float = read_f64(&src.join("threshold.f64"))[0];
    let methods = [(SolutionMethod::Jacobi, "jacobi"), (SolutionMethod::BiCGSTAB, "bicgstab"), (SolutionMethod::Multigrid, "multigrid")];
    let preconds = [(PreconditionMethod::None, "none"), (PreconditionMethod::Jacobi, "jacobi")];
    for (method, mname) in methods {
        for (pre, pname) in preconds {
            let mut x = DVector::from_vec(x0.clone());
            let result = catch_unwind(AssertUnwindSafe(|| {
                iterative_solve(&a, &b, &mut x, 50, method, 0.5, threshold, pre);
            }));
            let tag = format!("{mname}_pre_{pname}");
            match result {
                Ok(()) => {
                    write_text(&dir.join(format!("{tag}_status.txt")), "ok");
                    write_f64(&dir.join(format!("{tag}_x.f64")), x.as_slice());
                }
                Err(e) => write_text(&dir.join(format!("{tag}_status.txt")), &panic_text(e)),
            }
        }
    }
    // the reference's own unit test carries x from the Jacobi stage into the BiCGSTAB stage (linear_algebra.rs:340-378)
    let mut x = DVector::from_vec(x0.clone());
    let chained = catch_unwind(AssertUnwindSafe(|| {
        iterative_solve(&a, &b, &mut x, 50, SolutionMethod::Jacobi, 0.5, threshold, PreconditionMethod::Jacobi);
        let after_jacobi = x.clone();
        iterative_solve(&a, &b, &mut x, 50, SolutionMethod::BiCGSTAB, 0.5, threshold, PreconditionMethod::Jacobi);
        after_jacobi
    }));
    if let Ok(after_jacobi) = chained {
        write_f64(&dir.join("chained_x_after_jacobi.f64"), after_jacobi.as_slice());
        write_f64(&dir.join("chained_x_after_bicgstab.f64"), x.as_slice());
    }
    println!("iterative_solve dumped: {name}");
}

fn dump_initialize_flow(case: &str, mesh_file: &str, bcs: fn(&mut Mesh), inputs: &Path, out: &Path) {
    let dir = out.join(case);
    fs::create_dir_all(&dir).unwrap();
    let mut mesh = read_mesh(inputs.join("meshes").join(mesh_file).to_str().unwrap());
    bcs(&mut mesh);
    let result = catch_unwind(AssertUnwindSafe(|| initialize_flow(&mesh, 1e-3, 1000.0, 40)));
    match result {
        Ok((u, v, w, p)) => {
            write_text(&dir.join("initialize_flow_status.txt"), "ok");
            write_f64(&dir.join("initialize_flow_u.f64"), u.as_slice());
            write_f64(&dir.join("initialize_flow_v.f64"), v.as_slice());
            write_f64(&dir.join("initialize_flow_w.f64"), w.as_slice());
            write_f64(&dir.join("initialize_flow_p.f64"), p.as_slice());
        }
        Err(e) => write_text(&dir.join("initialize_flow_status.txt"), &panic_text(e)),
    }
}

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: dump_golden <inputs dir written by rust/export_inputs.py> <output dir>");
        std::process::exit(2);
    }
    let inputs = PathBuf::from(&args[1]);
    let out = PathBuf::from(&args[2]);
    fs::create_dir_all(&out).unwrap();
    // panics are results here, not noise
    std::panic::set_hook(Box::new(|_| {}));

    // ---- assembly (tests/golden/make_golden.py's cases)
    dump_assembly("3x3_cube", "3x3_cube.msh", cube_bcs, TVD_UMIST, &inputs, &out);
    dump_assembly("3x3_cube_mixed", "3x3_cube.msh", cube_bcs_mixed, MomentumDiscretization::CD1, &inputs, &out);
    dump_assembly("channel_flow", "channel_flow.msh", channel_bcs, MomentumDiscretization::CD1, &inputs, &out);

    // ---- solve_steady: the default stack (Multigrid + Jacobi preconditioner, lib.rs:58-86), BiCGSTAB and Jacobi solvers
    for (case, mesh_file, bcs) in [
        ("channel_flow", "channel_flow.msh", channel_bcs as fn(&mut Mesh)),
        ("3x3_cube", "3x3_cube.msh", cube_bcs as fn(&mut Mesh)),
    ] {
        dump_solve_steady(case, "multigrid_cd1_5it", mesh_file, bcs, settings_with(MomentumDiscretization::CD1, SolutionMethod::Multigrid, 50), 5, &inputs, &out);
        dump_solve_steady(case, "multigrid_umist_5it", mesh_file, bcs, settings_with(TVD_UMIST, SolutionMethod::Multigrid, 50), 5, &inputs, &out);
        dump_solve_steady(case, "multigrid_umist_20inner_5it", mesh_file, bcs, settings_with(TVD_UMIST, SolutionMethod::Multigrid, 20), 5, &inputs, &out);
        dump_solve_steady(case, "bicgstab_umist_5it", mesh_file, bcs, settings_with(TVD_UMIST, SolutionMethod::BiCGSTAB, 50), 5, &inputs, &out);
        dump_solve_steady(case, "jacobi_ud_5it", mesh_file, bcs, settings_with(MomentumDiscretization::UD, SolutionMethod::Jacobi, 50), 5, &inputs, &out);
    }
    dump_initialize_flow("channel_flow", "channel_flow.msh", channel_bcs, &inputs, &out);

    // ---- iterative_solve on exported systems
    let systems_dir = inputs.join("systems");
    if let Ok(entries) = fs::read_dir(&systems_dir) {
        let mut names: Vec<String> = entries.filter_map(|e| e.ok()).map(|e| e.file_name().to_string_lossy().to_string()).collect();
        names.sort();
        for name in names {
            dump_iterative_solve(&name, &inputs, &out);
        }
    }
    println!("done: {}", out.display());
}
