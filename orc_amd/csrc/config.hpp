// config.hpp — every environment switch of liborc_amd.so, in ONE place.
//
// Read once (orc_init) and again on orc_reload_environment() — never inside a solve: nothing on the hot path calls getenv, and two ranks
// of one job cannot drift apart because one of them changed its environment between iterations.  The tests flip switches inside one
// process: their `monkeypatch` (tests/conftest.py) calls orc_reload_environment() after every change.
// INTEGRATION.md lists these with their meaning; none changes a result (each group's tests compare its forms bit for bit).
#pragma once
#include <string>

namespace orc {

struct Config {
    // ---- schedule of a SIMPLE iteration (assembly.hip)
    bool triple_momentum = true;        // ORC_TRIPLE_MOMENTUM: u, v, w in lock-step on their shared pattern (0: one system per solve)
    bool concurrent_momentum = true;    // ORC_CONCURRENT_MOMENTUM: momentum set-ups / solves on streams and threads of their own
    bool two_stream_multigrid = true;   // ORC_TWO_STREAM_MULTIGRID: the set-up of level l+1 beside the smoothing of level l (one-system arm)
    bool early_p_hierarchy = true;      // ORC_EARLY_P_HIERARCHY: the p' hierarchy is built beside the momentum solves
    int stream_priorities = 3;          // ORC_STREAM_PRIORITIES: 3 solve class above set-up class, 2 the reverse, 1 by lane, 0 none (runtime.cpp)
    bool halo_overlap = true;           // ORC_HALO_OVERLAP: level-0 products of a partitioned mesh run their interior rows beside the exchange
    // ---- hierarchy set-up (amg.hip)
    bool amg_da = true;                 // ORC_AMG_DA: pairing by deferred acceptance (0: the lock-step rounds only — the fallback)
    int amg_da_steps = 1 << 22;         // ORC_AMG_DA_STEPS: proposals per chain before it is cut (test hook: the fallback finishes the job)
    int amg_da_group = 0;               // ORC_AMG_DA_GROUP: lanes per chain (0: by row length)
    bool amg_sibling = true;            // ORC_AMG_SIBLING: v and w try u's fine-level pairing first
    bool amg_shared_scaling = true;     // ORC_AMG_SHARED_SCALING: a level's two smoothing solves share one inverse diagonal and one set of scaled values
    bool amg_shared_galerkin = true;    // ORC_AMG_SHARED_GALERKIN: one symbolic Galerkin pass for u, v, w when their pairings agree
    bool amg_l0_mirror = true;          // ORC_AMG_L0_MIRROR: row-contiguous mirror of the fine level for the set-up's row walks
    std::string galerkin_groups;        // ORC_GALERKIN_GROUPS: lanes per coarse row by tier, "16,16,32,64" (test hook: every merge width)
    // ---- products (linalg.hip)
    int spmv_nt = -1;                   // ORC_SPMV_NT: non-temporal matrix loads 0 never / 1 always / -1 above 128 MB of stream
    bool spmv_narrow_cols = true;       // ORC_SPMV_NARROW_COLS: 16-bit column offsets on levels 0-1
    int materialize_scaling = 4;        // ORC_MATERIALIZE_SCALING: BiCGSTAB iterations from which a solve materialises its Jacobi-scaled values (0: never)
    int spmv_xwin_min_nnz = 24;         // ORC_SPMV_XWIN_MIN_NNZ: entries per row from which a coarse level gets the packed mirror + LDS windows (< 0: never)
    int spmv_grid = 0;                  // ORC_SPMV_GRID: cap of the product grids (test hook of the partial-sum bound; 0: kMaxGrid)
    int xwin_wgs_per_cu = 8;            // ORC_XWIN_WGS_PER_CU: window product workgroups per CU (test hook of the same bound)
    int xwin_cap = 0;                   // ORC_XWIN_CAP: window entries per block (test hook: forces the no-window fallback; 0: kXWinCap)
    int xwin_bitwords = 0;              // ORC_XWIN_BITWORDS / ORC_XWIN_SMALL_BITWORDS: bitmap spans of the window build (test hooks; 0: compiled sizes)
    int xwin_small_bitwords = 0;
    bool xwin_wg_per_block = true;      // ORC_XWIN_WG_PER_BLOCK: window products launch one workgroup per 256-row block and fold their sums inside the launch (0: 2 048 persistent workgroups, r04)
    bool xwin_level_cap = true;         // ORC_XWIN_LEVEL_CAP: a level's products size their LDS window to what the level needs (0: the compiled 40 KB, as until r04)
    // ---- Gauss-Seidel extension (gs.hip)
    bool gs_slotspace = true;           // ORC_GS_SLOTSPACE: GS-preconditioned BiCGSTAB in colour-sorted slot space (0: row space, as partitioned runs use)
    // ---- diagnostics (stderr; none touches a result)
    bool trace = false;                 // ORC_DEBUG_TRACE: progress markers of the SIMPLE driver and the solves
    bool amg_trace = false;             // ORC_AMG_TRACE: per-aggregation statistics and phase times (drains streams)
    bool arena_trace = false;           // ORC_ARENA_TRACE
    bool debug_nan = false;             // ORC_DEBUG_NAN: NaN / magnitude of the systems' fields after every solve (sequential schedule)
    bool debug_xwin = false;            // ORC_DEBUG_XWIN / ORC_XWIN_STATS: window statistics per level
    int debug_sync = 0;                 // ORC_DEBUG_SYNC: bit mask of steps of the lock-step level 1 after which the library stream is drained
    std::string inject_lane_error;      // ORC_DEBUG_INJECT_LANE_ERROR "rank:lane" (test hook: a failing set-up thread)
    bool keep_priority_classes = false; // ORC_DEBUG_KEEP_PRIORITY_CLASSES: classes although ranks share a card (scripts/archive/gpu_r05_b.sh: reproduces r04's stall)
};

const Config &cfg();   // the values of the last (re)load
void config_reload();  // re-reads the environment

}  // namespace orc
