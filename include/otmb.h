/*
 * otmb.h -- C ABI of libotmb_hip.so: the MI355X (gfx950) implementation of the sparse
 * transport-operator assembly path of TMIP-code/OceanTransportMatrixBuilder.jl v0.8.3.
 *
 * The reference is pure Julia and has no FFI layer; its
 * boundary for this path is the keyword-argument API exported at
 * src/OceanTransportMatrixBuilder.jl:31-36.  Each entry point below names the reference
 * function it replaces; julia/OceanTransportMatrixBuilderAMD.jl binds them with `ccall`
 * behind the reference's own function names.
 *
 * Conventions
 *  - Arrays are Julia's: column-major (nx,ny,nz), i fastest; Float64 values; Int64 indices,
 *    1-based exactly as Julia stores them (colptr, rowval, Lwet, Lwet3D).  `missing` in
 *    Lwet3D is 0; `missing`/`nothing` in Float64 inputs is NaN.
 *  - `_dev` entry points take DEVICE pointers and enqueue on the context's stream; the
 *    entry points without the suffix take HOST pointers (what Julia's ccall passes) and
 *    stage through device memory owned by the context.
 *  - Every function returns an otmb_status; otmb_last_error(ctx) gives the message, which
 *    for the reference's own failures is the reference's exact error string.
 *  - A context is bound to one GPU, is not shared between threads, and has no global state.
 */
#ifndef OTMB_H
#define OTMB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct otmb_ctx otmb_ctx;

typedef enum {
    OTMB_OK = 0,
    OTMB_ERR_RHO_NAN = 1,          /* "ρ contains NaNs"        src/matrixbuilding.jl:233 */
    OTMB_ERR_TADV_NAN = 2,         /* "Tadv contains NaNs."    src/matrixbuilding.jl:39  */
    OTMB_ERR_TKH_NAN = 3,          /* "TκH contains NaNs."     src/matrixbuilding.jl:61  */
    OTMB_ERR_TKVML_NAN = 4,        /* "TκVML contains NaNs."   src/matrixbuilding.jl:90  */
    OTMB_ERR_TKVDEEP_NAN = 5,      /* "TκVdeep contains NaNs." src/matrixbuilding.jl:114 */
    OTMB_ERR_FLUX_INTO_LAND = 6,   /* reference throws from Lwet3D[nothing] / push!(…, missing) */
    OTMB_ERR_UNKNOWN_TOPOLOGY = 7, /* "Unknown grid type"      src/gridtopology.jl:111-116 */
    OTMB_ERR_ALL_MISSING = 8,      /* AssertionError           src/velocities.jl:199-200 */
    OTMB_ERR_ALLOC = 9,
    OTMB_ERR_HIP = 10,
    OTMB_ERR_INVALID_ARG = 11,
    OTMB_ERR_NO_PLAN = 12,         /* fill/fetch without a successful plan */
    OTMB_ERR_NONCANONICAL_INDICES = 13, /* Lwet3D is not what makeindices(v3D) returns */
    OTMB_ERR_CAPACITY = 14,
    OTMB_ERR_PUSH_MASK = 15,       /* args.push_mask does not describe args.phi / args.lwet3d (nothing was written) */
    OTMB_ERR_ASYMMETRIC_PATTERN = 16, /* lump_and_spray: Graphs.SimpleGraph's ArgumentError for a one-directional T pattern */
    OTMB_ERR_GIVEN_FOREIGN = 17    /* otmb_tm_args.given: an operator that is NOT what this library derives for these arguments was handed to an
                                    * entry point that cannot add it (the asynchronous and the multi-slab builds): use otmb_transportmatrix_plan[_dev] */
} otmb_status;

/* gridmetrics.gridtopology (src/gridtopology.jl:1-16) */
typedef enum { OTMB_BIPOLAR = 0, OTMB_TRIPOLAR = 1, OTMB_UNKNOWN_TOPOLOGY = 2 } otmb_topology;

/* order of the face-flux arrays: the fields of the NamedTuple returned at src/velocities.jl:245-252 */
enum { OTMB_EAST = 0, OTMB_WEST = 1, OTMB_NORTH = 2, OTMB_SOUTH = 3, OTMB_TOP = 4, OTMB_BOTTOM = 5 };
/* order of the per-direction (nx,ny) metric arrays edge_length_2D[dir], distance_to_neighbour_2D[dir] */
enum { OTMB_DIR_WEST = 0, OTMB_DIR_EAST = 1, OTMB_DIR_SOUTH = 2, OTMB_DIR_NORTH = 3 };
/* order of the five returned matrices: (; T, Tadv, TκH, TκVML, TκVdeep), src/matrixbuilding.jl:149 */
enum { OTMB_T = 0, OTMB_TADV = 1, OTMB_TKH = 2, OTMB_TKVML = 3, OTMB_TKVDEEP = 4 };

/* ---- context ------------------------------------------------------------------------------ */
int32_t otmb_ctx_create(int32_t device_id, otmb_ctx **out);
void otmb_ctx_destroy(otmb_ctx *ctx);
/* Borrow a hipStream_t (e.g. torch's current stream) for every _dev call; NULL = the ctx's own. */
int32_t otmb_ctx_set_stream(otmb_ctx *ctx, void *hip_stream);
/* Enqueue on the device's DEFAULT (null) stream instead -- the stream a framework uses when it has not been given
 * another one (torch's default stream has handle 0, which otmb_ctx_set_stream reads as "the ctx's own"): the
 * library's kernels are then ordered with the caller's own kernels and copies on that stream.                  */
int32_t otmb_ctx_use_default_stream(otmb_ctx *ctx);
int32_t otmb_ctx_synchronize(otmb_ctx *ctx);
/* Host-pointer entry points only.  The grid (gridmetrics, indices) does not change between the time slices a TMIP script
 * loops over (README.md:65-80), but every call hands the same host arrays over again: with reuse_grid on, a
 * grid-constant array (v3D, thkcello, Lwet3D, Lwet, wet3D, the 2-D metrics, zt) whose host pointer and size equal those
 * of the previous upload into its staging slot is NOT copied again -- the caller promises not to have modified it in
 * between.  Off by default (every call uploads everything, like the reference reads everything).                   */
int32_t otmb_ctx_set_reuse_grid(otmb_ctx *ctx, int32_t on);
/* Host-pointer entry points only.  otmb_facefluxes leaves the six ϕ arrays in the context's device staging; with
 * reuse_fluxes on, an otmb_transportmatrix_plan that is handed those very host arrays back (same pointers, same size)
 * does not upload them again (259 MB at 1 degree) -- the caller promises not to have modified them in between, which
 * is what a script that passes facefluxesfrommasstransport's result straight to transportmatrix does (README.md:65-80).
 * Off by default.                                                                                                */
int32_t otmb_ctx_set_reuse_fluxes(otmb_ctx *ctx, int32_t on);
/* reuse_grid and reuse_fluxes are independent promises (either may be on without the other).  Diagnostics: the bytes the
 * host-pointer entry points of this context have copied to the device so far -- what the two flags save.            */
int64_t otmb_ctx_uploaded_bytes(const otmb_ctx *ctx);
/* Pinned (page-locked) host memory owned by the context, for the arrays a caller hands to the host-pointer entry points:
 * a buffer inside such a block is the DMA's own source / target -- no staging copy, no page faults on freshly allocated
 * output arrays (1 GB of them per transportmatrix at 1 degree).  Julia: unsafe_wrap(Array, ptr, n) + a finalizer calling
 * otmb_host_free.  Freed blocks are kept and handed out again (up to 4 GiB idle).  ctx may be NULL (a caller that only uses an
 * otmb_mgpu has no single-device context): the pool belongs to no context.                                          */
int32_t otmb_host_alloc(otmb_ctx *ctx, int64_t bytes, void **out);
/* Lifetime and threads: the blocks belong to ONE pool of the process, behind a lock, which no context owns: otmb_ctx_destroy
 * frees none of them, and otmb_host_free IGNORES its context argument (NULL and an already destroyed context are fine) and may
 * be called from any thread at any time -- a garbage collector's finalizer thread while another thread is inside a call on the
 * context, or a finalizer that runs after the atexit hook that destroyed the context (Julia runs atexit hooks BEFORE its final
 * finalizer sweep).  Returns OTMB_ERR_INVALID_ARG for a pointer that is not a live block (e.g. freed twice).              */
int32_t otmb_host_free(otmb_ctx *ctx_ignored, void *p);
/* blocks handed out and not yet freed, their bytes, and the bytes kept idle for reuse (any argument may be NULL) */
int32_t otmb_host_pool_stats(int64_t *blocks_in_use, int64_t *bytes_in_use, int64_t *bytes_idle);
/* Speed only, never results: the order in which the fill pass of transportmatrix takes its tiles of 256 columns.
 * rows_per_band = 0: ascending wet rank (i, then j, then k).  R > 0: MARCH order -- the tiles of a band of R grid rows
 * are taken level after level before the next band starts, so that the levels above / below a tile (the vertical
 * neighbours of src/matrixbuilding.jl:280-296, :450-477) were read moments ago and are still in the L2 / Infinity Cache
 * instead of one whole level (124 MB of inputs on a 0.25 degree grid) earlier.  -1 (default): the library's choice (bands of
 * 8 rows: with the matrices written by non-temporal stores, 8 % faster than wet-rank order at 1 and at 0.25 degree).
 * On grids whose rows are longer than 1536 cells a band is further cut into equal blocks of columns (3 x 1200 on a 3600-cell
 * row): what an XCD's L2 keeps from one level to the next is rows x columns cells -- HBM fetch of the fill pass on the 0.1 degree
 * grid 70 -> 48 GB, its time unchanged (profiles/r04/README.md section 10).  Environment, read when a context is created, for
 * experiments only: OTMB_MARCH_ROWS, OTMB_MARCH_COLS (0 = whole rows).                                                       */
int32_t otmb_ctx_set_tile_order(otmb_ctx *ctx, int32_t rows_per_band);
const char *otmb_last_error(const otmb_ctx *ctx);
const char *otmb_status_string(int32_t status); /* the reference's error text for codes 1-8 */
const char *otmb_version(void);
/* Optional per-kernel timing: when enabled every kernel launch is bracketed by two hipEvents on
 * the launch stream.  collect() synchronises, adds the elapsed times per kernel id, returns the
 * sums (ms) and launch counts since the previous collect for ids 0..n-1 and resets them.      */
int32_t otmb_ctx_timing_enable(otmb_ctx *ctx, int32_t on);
int32_t otmb_ctx_timing_collect(otmb_ctx *ctx, double *ms_sum, int64_t *count, int32_t n);
const char *otmb_kernel_name(int32_t kernel_id);
/* Diagnostic (not part of any result): what the box this context runs on sustains for one plain HBM read stream and one plain
 * non-temporal HBM write stream (2 GiB each, all CUs, 16 bytes per lane, eight accesses in flight), in GB/s.  bench.py reports a
 * kernel's time as a fraction of the rate these two give for the kernel's own read : write mix.  Allocates 2 GiB for the call.   */
int32_t otmb_ctx_box_ceilings(otmb_ctx *ctx, double *read_gbs, double *write_gbs);
/* Diagnostic, DESTRUCTIVE for the outputs: the plainest kernel there is over the very arrays a kernel reads and writes -- every byte of the
 * n_in input arrays read once, every byte of the n_out output arrays written once (device pointers, 16-byte aligned; at most 24 each), cut
 * into `tiles` proportional slices taken by one workgroup each in the fill pass's XCD-contiguous order.  *gbs = bytes / time: what an ideal
 * streaming kernel with this byte mix reaches ON THESE ALLOCATIONS -- the fill pass's time depends reproducibly on where its ~30 arrays lie
 * in the HBM (profiles/r04 section 12), which the two plain streams of otmb_ctx_box_ceilings do not see.                              */
int32_t otmb_ctx_stream_mix(otmb_ctx *ctx, int32_t n_in, const void *const *in, const int64_t *in_bytes, int32_t n_out,
                            void *const *out, const int64_t *out_bytes, int64_t tiles, double *gbs);

/* ---- makeindices(v3D)  -- src/matrixbuilding.jl:10-24 ------------------------------------- *
 * wet = !isnan(v3D).  Outputs (any may be NULL): lwet3d (nx*ny*nz) wet rank or 0;
 * lwet (capacity nx*ny*nz; first N valid) ascending 1-based linear indices of wet cells;
 * wet3d (nx*ny*nz bytes, 0/1).  *n_wet receives N (host memory in both variants).        */
int32_t otmb_makeindices_dev(otmb_ctx *ctx, const double *v3d, int64_t nx, int64_t ny, int64_t nz,
                             int64_t *lwet3d, int64_t *lwet, uint8_t *wet3d, int64_t *n_wet);
int32_t otmb_makeindices(otmb_ctx *ctx, const double *v3d, int64_t nx, int64_t ny, int64_t nz,
                         int64_t *lwet3d, int64_t *lwet, uint8_t *wet3d, int64_t *n_wet);

/* ---- facefluxes(umo, vmo, gridmetrics, indices; FillValue) -- src/velocities.jl:190-255,
 *      including nofluxboundaries! (:154-179) and the Float64 conversion of
 *      facefluxesfrommasstransport (:118-130).
 * umo/vmo: (nx,ny,nz), Float64 (src_is_f32 == 0) or Float32 (== 1, the CMIP on-disk type);
 * they are NOT modified (the reference mutates only its converted copies).  `fill` is the
 * _FillValue promoted to Float64.  phi[6]: outputs in OTMB_EAST..OTMB_BOTTOM order.         */
int32_t otmb_facefluxes_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                            const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                            int32_t topology, double *const phi[6]);
int32_t otmb_facefluxes(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                        const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                        int32_t topology, double *const phi[6]);

/* Depth-slab variant for multi-GPU runs (no counterpart in the single-process reference; it continues
 * the bottom-up continuity recurrence of src/velocities.jl:236-243 across slabs without re-association):
 * the nz levels handed in are levels [k0,k1) of a deeper grid and top_below (nx*ny, NULL for the deepest
 * slab) is ϕtop of level k1 from the slab below.  Asynchronous; the :199-200 assertion concerns the
 * whole grid, so each slab's two validity flags are read with otmb_facefluxes_slab_flags (synchronises)
 * and OR-ed across slabs by the caller.                                                             */
int32_t otmb_facefluxes_slab_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                 const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                                 int32_t topology, double *const phi[6], const double *top_below, uint16_t *push_mask);
/* The same for ONE ROW BAND of the plane: rows [j0, j1) (0-based) of every level.  A cell's fluxes depend on its own column's
 * top_below and on INPUTS of its west / south neighbours only, so the chain over depth slabs can run piece by piece -- slab s piece c
 * waits for slab s + 1 piece c, not for its whole plane (src/velocities.jl:236-243) -- with bit-identical arrays for any
 * partition of the rows.  top_below / the phi arrays are whole-plane arrays as above; first = 1 on the first piece of a field
 * (all pieces of a field share one pair of validity flags).                                                               */
int32_t otmb_facefluxes_rows_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                 const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                                 int32_t topology, double *const phi[6], const double *top_below, uint16_t *push_mask,
                                 int64_t j0, int64_t j1, int32_t first);
int32_t otmb_facefluxes_slab_flags(otmb_ctx *ctx, int32_t *u_valid, int32_t *v_valid);
/* Speed only.  nofluxboundaries! (src/velocities.jl:161-175) looks at the wet byte of every cell AND of its east, west,
 * south and north (or fold) neighbours, for every level of every time slice -- all grid constants.  otmb_wetflags_dev folds
 * the five into one byte per cell (bit 0 the cell, bits 1-4 east, west, south, north; asynchronous; once per grid), and
 * otmb_facefluxes_flags_dev is otmb_facefluxes_slab_dev reading that array instead of wet3d: five loads per cell and
 * level instead of nine, the same six ϕ arrays and push mask bit for bit.                                      */
int32_t otmb_wetflags_dev(otmb_ctx *ctx, const uint8_t *wet3d, int64_t nx, int64_t ny, int64_t nz, int32_t topology,
                          uint8_t *wetflags);
int32_t otmb_facefluxes_flags_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                  const uint8_t *wetflags, double fill, int64_t nx, int64_t ny, int64_t nz,
                                  int32_t topology, double *const phi[6], const double *top_below, uint16_t *push_mask);
/* Speed only: facefluxes that ALSO counts.  Which rows the four operators of transportmatrix hold in column c
 * (src/matrixbuilding.jl:244-296, :348-415, :450-477) follows from the wet mask, the mixed-layer mask zt[k] < mlotst[i,j] (:85) and the
 * signs of the fluxes c's six neighbours push with -- which, for fluxes facefluxes itself writes, are c's OWN six fluxes
 * (ϕwest[E] = ϕeast[c], ..., ϕbottom[A] = ϕtop[c]: src/velocities.jl:206-224, :238-240).  otmb_facefluxes_counts_dev is
 * otmb_facefluxes_flags_dev whose kernel also accumulates, per tile of 256 matrix columns, the row counts of Tadv and TκVML that
 * transportmatrix's counting pass would derive (the counts of TκH, TκVdeep and T's reserved union depend on the wet mask alone and come
 * from the grid's tables); an otmb_transportmatrix_dev / _plan_dev that is then handed EXACTLY these ϕ arrays, this push_mask pointer,
 * mlotst, zt, lwet3d, n_wet, topology, upwind and only_t -- with no other facefluxes call on the context in between -- skips its
 * counting pass (6-7 % of a device-resident step).  Any other transportmatrix call counts for itself as before, and the fill pass
 * compares every tile's counts with the columns it builds (OTMB_ERR_PUSH_MASK), so stale counts cannot produce a wrong matrix.
 * In this mode push_mask is NOT written: the pointer only names the fluxes (the library knows and never reads it as a mask).  Whole
 * grids only (no top_below, wet ranks from 1).  Falls back to otmb_facefluxes_flags_dev's behaviour (mask written, no counts) when
 * nx < 3 or OTMB_COUNT_IN_FF=0.
 * tables: once per grid by otmb_count_tables_dev from the indices and otmb_wetflags_dev's bytes (otmb_count_tables_bytes bytes: the wet
 * rank of the first wet cell of every 64-cell segment a wave of the kernel covers, and the static counts of every tile).  Both
 * asynchronous.                                                                                                                   */
typedef struct {
    const void *tables;      /* otmb_count_tables_dev */
    const int64_t *lwet3d;   /* indices.Lwet3D */
    const double *mlotst;    /* (nx,ny) */
    const double *zt;        /* (nz) */
    int64_t n_wet;
    int32_t upwind;          /* as otmb_tm_args.upwind */
    int32_t only_t;          /* as otmb_tm_args.only_t */
} otmb_ff_counts;
int64_t otmb_count_tables_bytes(const otmb_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t n_wet);
int32_t otmb_count_tables_dev(otmb_ctx *ctx, const int64_t *lwet3d, const int64_t *lwet, const uint8_t *wetflags, int64_t n_wet,
                              int64_t nx, int64_t ny, int64_t nz, int32_t topology, void *tables);
int32_t otmb_facefluxes_counts_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                   const uint8_t *wetflags, double fill, int64_t nx, int64_t ny, int64_t nz,
                                   int32_t topology, double *const phi[6], uint16_t *push_mask, const otmb_ff_counts *counts);
/* The same for a depth slab (otmb_facefluxes_slab_dev / _rows_dev: SURVEY 8e, the chain over depth slabs).  The slab's nz levels are
 * levels [k_own0, k_own0 + nz) of its EXTENDED local grid of nz_ext levels -- the owned levels plus one halo level above (k_own0 = 1)
 * and / or below (k_own0 + nz < nz_ext), whose cells are neighbours of owned cells and rows of the slab's matrices but not columns.
 * wetflags (otmb_wetflags_dev of the extended grid's mask), counts->zt and counts->lwet3d are arrays of the EXTENDED grid, exactly the
 * ones the slab's otmb_tm_args will name; umo, vmo, phi[6], push_mask and top_below hold the owned levels only, as for
 * otmb_facefluxes_rows_dev.  counts->n_wet: the slab's owned wet cells; wet_base: the global 0-based wet rank of the first of them
 * (otmb_transportmatrix_set_slab).  The transportmatrix call that may skip its counting pass is the one whose arguments are the extended
 * arrays these owned-level pointers lie in (phi[d] - k_own0 * nx * ny, ...), after the halo planes' fluxes have been put in place.
 * j0, j1, first: a row band, as for otmb_facefluxes_rows_dev (bands other than the whole plane count only under the four-row wave
 * geometry of large planes; otherwise -- and wherever otmb_facefluxes_counts_dev falls back -- the mask is written and nothing is
 * counted).                                                                                                                       */
typedef struct {
    int64_t k_own0;   /* first owned level inside the extended local grid (0-based): 0 or 1 */
    int64_t nz_ext;   /* levels of the extended local grid (<= nz + 2) */
    int64_t wet_base; /* global 0-based wet rank of the slab's first owned wet cell */
} otmb_ff_slab;
int32_t otmb_count_tables_slab_dev(otmb_ctx *ctx, const int64_t *lwet3d, const int64_t *lwet, const uint8_t *wetflags, int64_t n_wet,
                                   int64_t nx, int64_t ny, int64_t nz, int32_t topology, const otmb_ff_slab *slab, void *tables);
int32_t otmb_facefluxes_slab_counts_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wetflags,
                                        double fill, int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6],
                                        const double *top_below, uint16_t *push_mask, const otmb_ff_counts *counts,
                                        const otmb_ff_slab *slab, int64_t j0, int64_t j1, int32_t first);
/* 1 when the last facefluxes call on this context counted (its push_mask argument was therefore not written, and need not be completed
 * for halo planes by otmb_push_mask_dev), 0 when it took the plain kernel or the counts have been consumed.  No device work.   */
int32_t otmb_facefluxes_counts_pending(const otmb_ctx *ctx);
/* The same two flags for EVERY facefluxes call on this context since the previous call of this function, oldest
 * first (a pipeline of asynchronous steps: the reference asserts per call, src/velocities.jl:199-200, so a field
 * without a single valid value in step 3 of 12 must not be hidden by steps 4-12).  Synchronises.  At most the 64
 * most recent calls are remembered: drain the pipeline at least that often.                                  */
int32_t otmb_facefluxes_pending_flags(otmb_ctx *ctx, int32_t capacity, int32_t *u_valid, int32_t *v_valid,
                                      int32_t *n_calls);

/* Push mask: what the pattern of the advection operator needs to know about a cell's six fluxes, 16 bits per
 * cell.  Bits 0-5: the cell sends mass through its west, east, south, north, bottom, top face under upwind
 * weighting, i.e. max(ϕwest,0), min(ϕeast,0), max(ϕsouth,0), min(ϕnorth,0), max(ϕbottom,0), min(ϕtop,0) is
 * non-zero (src/matrixbuilding.jl:244,253,262,271,280,289); bit 6: the cell is wet; bits 8-14: the same for
 * centred weighting (ϕ/2 non-zero).  otmb_facefluxes_slab_dev writes it for the cells it computes when
 * push_mask is not NULL (the fluxes are in registers there); otmb_push_mask_dev derives it from existing ϕ
 * arrays for the cells [first, first+count) (0-based linear indices; halo planes of a depth slab, or fluxes
 * that were modified after facefluxes).  Asynchronous.                                                  */
int32_t otmb_push_mask_dev(otmb_ctx *ctx, const double *const phi[6], const int64_t *lwet3d, int64_t first,
                           int64_t count, uint16_t *push_mask);

/* ---- velocity2fluxes / fluxes2velocity -- src/velocities.jl:10-39, :50-74 (nanmean2 :89-93, nanmin2 :108);
 *      with facefluxes they make facefluxesfromvelocities (:140-151).  Default C-grid (u on east faces, v on
 *      north faces; the reference passes C-grid fields through, src/gridcellgeometry.jl:104).
 * u/v (or phi_i/phi_j): (nx,ny,nz) Float64 or Float32 (src_is_f32); rho: (nx,ny,nz) or NULL with rho_scalar;
 * edge_east/edge_north: gridmetrics.edge_length_2D[:east], [:north] (nx,ny).  Every cell is computed, as in
 * the reference.  A bipolar topology is an error (the reference indexes thkcello[nothing] at j == ny).    */
/* B-grid (u, v on the NE corner, e.g. ACCESS-ESM1-5 uo/vo) -> default C-grid:
 * interpolateontodefaultCgrid(…, ::BGridCell), src/gridcellgeometry.jl:106-140: _FillValue -> 0, then
 * u2 = 0.5 (u + u one row south), v2 = 0.5 (v + v one cell west), zero shifted in at j == 1 / i == 1.    */
int32_t otmb_bgrid_to_cgrid_dev(otmb_ctx *ctx, const void *u, const void *v, int32_t src_is_f32, double fill,
                                int64_t nx, int64_t ny, int64_t nz, double *u2, double *v2);
int32_t otmb_bgrid_to_cgrid(otmb_ctx *ctx, const void *u, const void *v, int32_t src_is_f32, double fill,
                            int64_t nx, int64_t ny, int64_t nz, double *u2, double *v2);
int32_t otmb_velocity2fluxes_dev(otmb_ctx *ctx, const void *u, const void *v, int32_t src_is_f32, const double *rho,
                                 double rho_scalar, const double *thkcello, const double *edge_east,
                                 const double *edge_north, int64_t nx, int64_t ny, int64_t nz, int32_t topology,
                                 double *phi_i, double *phi_j);
int32_t otmb_fluxes2velocity_dev(otmb_ctx *ctx, const void *phi_i, const void *phi_j, int32_t src_is_f32,
                                 const double *rho, double rho_scalar, const double *thkcello, const double *edge_east,
                                 const double *edge_north, int64_t nx, int64_t ny, int64_t nz, int32_t topology,
                                 double *u, double *v);
int32_t otmb_velocity2fluxes(otmb_ctx *ctx, const void *u, const void *v, int32_t src_is_f32, const double *rho,
                             double rho_scalar, const double *thkcello, const double *edge_east,
                             const double *edge_north, int64_t nx, int64_t ny, int64_t nz, int32_t topology,
                             double *phi_i, double *phi_j);
int32_t otmb_fluxes2velocity(otmb_ctx *ctx, const void *phi_i, const void *phi_j, int32_t src_is_f32, const double *rho,
                             double rho_scalar, const double *thkcello, const double *edge_east,
                             const double *edge_north, int64_t nx, int64_t ny, int64_t nz, int32_t topology,
                             double *u, double *v);

/* ---- transportmatrix(; ϕ, mlotst, gridmetrics, indices, ρ, κH, κVML, κVdeep, upwind)
 *      -- src/matrixbuilding.jl:128-150 with buildTadv/TκH/TκVML/TκVdeep (:31-120), the three
 *      *_operator_sparse_entries generators (:221-299, :337-418, :438-479), sparse() x4 and
 *      T = Tadv + TκH + TκVML + TκVdeep (:147), fused: each wet cell's column of all five
 *      matrices is produced directly in CSC order.                                          */
/* A SparseMatrixCSC{Float64,Int64} by its three arrays, 1-based as Julia stores them: colptr (columns + 1), rowval / nzval (nnz). */
typedef struct {
    const int64_t *colptr;
    const int64_t *rowval;
    const double *nzval;
    int64_t nnz;
} otmb_csc;

typedef struct {
    int64_t nx, ny, nz;
    int32_t topology;            /* otmb_topology */
    int32_t upwind;              /* 1: upwind (default), 0: centred */
    int64_t n_wet;               /* indices.N */
    const double *phi[6];        /* ϕ.east, west, north, south, top, bottom: (nx,ny,nz) */
    const double *v3d;           /* gridmetrics.v3D      (nx,ny,nz), NaN on land */
    const double *thkcello;      /* gridmetrics.thkcello (nx,ny,nz) */
    const double *rho;           /* ρ as (nx,ny,nz) array, or NULL to use rho_scalar */
    double rho_scalar;
    const int64_t *lwet3d;       /* indices.Lwet3D (nx,ny,nz), 0 = missing */
    const int64_t *lwet;         /* indices.Lwet (n_wet): ascending 1-based linear indices of the wet cells */
    const double *edge_length[4];/* gridmetrics.edge_length_2D[dir]           (nx,ny), OTMB_DIR_* order */
    const double *dist_nbr[4];   /* gridmetrics.distance_to_neighbour_2D[dir] (nx,ny) */
    const double *area2d;        /* gridmetrics.area2D (nx,ny) */
    const double *zt;            /* gridmetrics.zt (nz) */
    const double *mlotst;        /* mlotst (nx,ny), NaN = missing */
    double kappa_h, kappa_vml, kappa_vdeep;
    const uint16_t *push_mask;   /* optional (device pointer, NULL = derived from phi and lwet3d by the library): the
                                  * per-cell push mask written by otmb_facefluxes_slab_dev / otmb_push_mask_dev for
                                  * exactly these phi and this wet mask; lets the counting pass read 2 bytes per
                                  * neighbour instead of re-reading the six ϕ arrays.  Host-pointer entry points
                                  * ignore it.                                                                   */
    int32_t only_t;              /* extension (0 = the reference's behaviour): non-zero builds T alone -- the four operator
                                  * matrices are still evaluated (T is their sum) but neither counted nor written: their
                                  * nnz come out 0 and their output pointers may be NULL.  Halves the bytes written.      */
    int32_t ignore_ops;          /* bit m (m = OTMB_TADV .. OTMB_TKVDEEP): the caller already HAS operator m (transportmatrix's
                                  * Tadv / TκH / TκVML / TκVdeep keywords, src/matrixbuilding.jl:133-143) -- the reference then never
                                  * builds it, so nothing it alone would have raised is raised: "Tadv contains NaNs.", "ρ contains
                                  * NaNs" and a flux into land for Tadv, "TκH / TκVML / TκVdeep contains NaNs." for the others.  The
                                  * matrices of ignored operators and T are still written and are the caller's to discard.       */
    int32_t skip_ops;            /* extension: bit m (m = OTMB_T .. OTMB_TKVDEEP): the caller does not WANT matrix m -- it is evaluated where T needs
                                  * it but neither counted, written nor copied home (its nnz comes out 0, its outputs may be NULL).  only_t is
                                  * skip_ops = the four operator bits.  buildTadv / buildTκH / buildTκVML / buildTκVdeep (src/matrixbuilding.jl:31-120)
                                  * are this call with every bit but one set, and ignore_ops for the operators they do not build.               */
    otmb_csc given[5];           /* given[m].colptr != NULL (m = OTMB_TADV .. OTMB_TKVDEEP; given[OTMB_T] stays zero): transportmatrix's
                                  * Tadv = / TκH = / TκVML = / TκVdeep = keyword (src/matrixbuilding.jl:133-143) -- the caller PASSES operator m
                                  * (N x N; device arrays for the _dev entry points, host arrays for the others).  As in the reference it
                                  * is then NOT built: not counted, not stored, not copied home (nnz[m] comes back 0, the outputs of m may be
                                  * NULL: the caller returns the very object it passed, :149), nothing it alone would raise is raised (as
                                  * ignore_ops), and T = ((Tadv + TκH) + TκVML) + TκVdeep (:147) is formed with the GIVEN matrix:
                                  *  - TκH / TκVdeep are functions of the grid and κ alone.  A given one whose three arrays are bit for bit
                                  *    what this library derives for these gridmetrics / indices / κ ("derived": checked on the device by
                                  *    one pass that compares instead of stores, the verdict kept per context and operator until one of
                                  *    the arrays it names changes -- otmb_ctx_forget_given) is re-derived in registers by the fill pass:
                                  *    its 16 nnz + 8 (N + 1) bytes are not written (29 % of the fill pass's bytes at 1 degree for
                                  *    TκH + TκVdeep, 40 % of the bytes a host caller waits for); TκH's values are read where they lie when
                                  *    that is cheaper than re-deriving them.
                                  *  - A given TκH / TκVdeep with exactly the derived ROWS and other values -- built with another κ, which is
                                  *    what a caller passes who leaves the call's own κH / κVdeep at their defaults -- is treated alike, and
                                  *    the fill pass READS its values where they lie: T carries the given values, in every protocol.
                                  *  - any other given matrix (another pattern, Tadv, TκVML) is "foreign": the built operators
                                  *    are written as usual and T is formed by the device sparse add (otmb_spadd_*_dev: left fold, exact
                                  *    zeros dropped) from the given arrays where they lie.  Two-phase entry points only
                                  *    (otmb_transportmatrix_plan[_dev] + fill / fetch: the plan's nnz[0] is then the sum of the four
                                  *    operands' counts); the asynchronous and multi-slab builds return OTMB_ERR_GIVEN_FOREIGN.
                                  * In a depth-slab launch (otmb_transportmatrix_set_slab) given[m] names the slab's columns: colptr
                                  * its n_wet + 1 entries (global numbering), rowval / nzval the entries from colptr[0] on, nnz their number. */
} otmb_tm_args;
/* The verdicts on otmb_tm_args.given are keyed to array ADDRESSES (the given matrix's and the gridmetrics / indices arrays') and κ.
 * A device-resident caller that rewrites one of those arrays in place calls this before the next transportmatrix; the host-pointer
 * entry points do it themselves whenever they upload such an array (i.e. always, unless otmb_ctx_set_reuse_grid promises otherwise). */
int32_t otmb_ctx_forget_given(otmb_ctx *ctx);
/* Diagnostics: how the last plan / _dev call on this context treated operator m: 0 not given, 1 given and derived, 2 given and foreign,
 * 3 given with the derived rows and other values (read by the fill pass). */
int32_t otmb_ctx_given_state(const otmb_ctx *ctx, int32_t m);
/* ... and how many comparing passes the context has run so far (a time loop with resident arrays runs ONE). */
int64_t otmb_ctx_given_checks(const otmb_ctx *ctx);

/* Two-phase protocol so the CALLER allocates the outputs (Julia owns its SparseMatrixCSC buffers).
 * plan: the nnz of the four operator matrices (exact: their patterns depend on the wet mask, the flux
 *   signs and the mixed-layer mask only) and an UPPER BOUND for T = the union pattern (T drops entries
 *   whose sum is exactly zero, src/matrixbuilding.jl:147 -- rare, and only known once values exist).
 * fill (synchronous): writes colptr[m] (n_wet+1), rowval[m], nzval[m] for the five matrices in
 *   OTMB_T..OTMB_TKVDEEP order, raises the reference's errors, compacts T if entries cancelled;
 *   otmb_transportmatrix_nnz then gives the final counts (nnz[0] <= the planned bound).
 * The args of the last plan are remembered by the context; a plan is consumed by its fill (fill again = plan again). */
int32_t otmb_transportmatrix_plan_dev(otmb_ctx *ctx, const otmb_tm_args *args, int64_t nnz[5]);
int32_t otmb_transportmatrix_fill_dev(otmb_ctx *ctx, int64_t *const colptr[5], int64_t *const rowval[5],
                                      double *const nzval[5]);
/* Asynchronous variant for device-resident callers that preallocate the outputs at an upper bound (a column
 * holds at most 7, 7, 5, 3, 3 entries of T, Tadv, TκH, TκVML, TκVdeep): count -> scan -> fill are enqueued back
 * to back with no host round trip.  capacity[m] = entries rowval[m]/nzval[m] can hold; colptr[m] holds
 * n_wet+1.  otmb_transportmatrix_result synchronises, raises the reference's errors / OTMB_ERR_CAPACITY,
 * compacts T if entries cancelled and returns the five nnz.
 * One exception to "no host round trip": the FIRST call for a grid (and the first after otmb_ctx_set_stream changed the stream)
 * builds the fill pass's tile order and waits for its number of heavy tiles -- one stream synchronisation per grid, not per step. */
int32_t otmb_transportmatrix_dev(otmb_ctx *ctx, const otmb_tm_args *args, int64_t *const colptr[5],
                                 int64_t *const rowval[5], double *const nzval[5], const int64_t capacity[5]);
/* Extension for device-resident pipelines that go from (umo, vmo) straight to the matrices and do not need the six ϕ arrays themselves:
 * ONE call = otmb_facefluxes_counts_dev + otmb_transportmatrix_dev with the SAME five matrices bit for bit, but only ϕtop is ever stored
 * (phi_top: nx*ny*nz doubles of the caller's; it holds facefluxes' ϕtop afterwards).  The other five fluxes are what facefluxes would have
 * stored -- ϕeast / ϕwest / ϕnorth / ϕsouth are masked copies of umo / vmo (nofluxboundaries! + replace + shift, src/velocities.jl:161-175,
 * :203-224), ϕbottom is ϕtop of the level below (:238-240) -- and the fill pass re-derives them where it uses them: 64 bytes per cell less
 * HBM traffic than writing six arrays and reading them back (~14 % of a step).  args->phi and args->push_mask are ignored; wetflags /
 * count_tables as for otmb_facefluxes_counts_dev; asynchronous like otmb_transportmatrix_dev (otmb_transportmatrix_result,
 * otmb_facefluxes_pending_flags collect the verdicts).  Whole grids, nx >= 3, nz <= 128.  NOT a replacement for the two-call API:
 * facefluxesfrommasstransport returns the six arrays (src/velocities.jl:245-254).                                                       */
int32_t otmb_step_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, double fill, const uint8_t *wetflags,
                      const void *count_tables, double *phi_top, const otmb_tm_args *args, int64_t *const colptr[5],
                      int64_t *const rowval[5], double *const nzval[5], const int64_t capacity[5]);
int32_t otmb_transportmatrix_result(otmb_ctx *ctx, int64_t nnz[5]);
/* Several otmb_transportmatrix_dev calls may be enqueued before one otmb_transportmatrix_result (a pipeline over time
 * slices).  Every call keeps its own error flags, its own nnz and its own output arrays: like every transportmatrix call of
 * the reference returns its own five matrices (src/matrixbuilding.jl:147-149).  result reports the FIRST call that
 * failed -- where the reference would have thrown (src/matrixbuilding.jl:39,61,90,114,233) -- with "(asynchronous step k
 * of n)" appended to otmb_last_error; when none failed it returns the nnz of the last call, and EVERY call's T has been
 * compacted in that call's own arrays if entries of it cancelled.  The outputs of a call are defined until a later call is
 * handed the same arrays (a caller that keeps one result set, like bench.py, keeps the last call's matrices).
 * failed_step: 0-based index of the failing call among the calls since the previous result, -1 if the last result found
 * no failure.  result_step (after result): status and nnz of the k-th of those calls.                              */
int32_t otmb_transportmatrix_failed_step(otmb_ctx *ctx, int64_t *step);
int32_t otmb_transportmatrix_result_step(otmb_ctx *ctx, int64_t step, int64_t nnz[5]);

/* Depth-slab partition (multi-GPU).  The wet index is k-slowest (src/matrixbuilding.jl:14-15), so a slab
 * of levels owns a contiguous column range of every matrix.  set_slab (before plan / _dev): the local
 * grid holds this rank's levels plus halo levels that act as neighbours only; args.lwet lists the OWNED
 * wet cells (local linear indices), args.n_wet their number, lwet3d holds GLOBAL wet ranks and wet_base
 * = (global rank of the first owned wet cell) - 1.  set_nnz_base (before fill / _dev): entries of each
 * matrix owned by the slabs above, so that colptr (n_wet+1 entries, local columns) is written with
 * global offsets and the slabs' arrays concatenate into the global CSC.  set_slab(ctx, 0) resets.   */
int32_t otmb_transportmatrix_set_slab(otmb_ctx *ctx, int64_t wet_base);
int32_t otmb_transportmatrix_set_nnz_base(otmb_ctx *ctx, const int64_t nnz_base[5]);
int32_t otmb_transportmatrix_plan(otmb_ctx *ctx, const otmb_tm_args *args, int64_t nnz[5]);
int32_t otmb_transportmatrix_nnz(otmb_ctx *ctx, int64_t nnz[5]);
/* host variant of fill: nnz_out receives the final counts (trim T's arrays to nnz_out[0]) */
int32_t otmb_transportmatrix_fetch(otmb_ctx *ctx, int64_t *const colptr[5], int64_t *const rowval[5],
                                   double *const nzval[5], int64_t nnz_out[5]);

/* ---- the same two calls over SEVERAL GPUs of one process: transportmatrix(...; devices = 0:7) ---------------------------
 * No counterpart in the single-threaded reference (src/matrixbuilding.jl:128-150 is the call shape that is kept).  The wet
 * index is k-slowest (src/matrixbuilding.jl:14-15): a slab of consecutive levels owns a contiguous column range of every matrix
 * and a contiguous range of every (nx,ny,nz) array.  An otmb_mgpu owns one context and one host thread per device, cuts the
 * levels with otmb_balanced_partition (wet counts as even as a sweep gets them) and moves every slab over ITS device's PCIe
 * link.  HOST pointers of the WHOLE grid go in and come out, exactly as for otmb_facefluxes / otmb_transportmatrix_plan /
 * _fetch, and the results are the same bit for bit:
 *   otmb_mgpu_facefluxes            the continuity recurrence (src/velocities.jl:236-243) is a chain in k with a fixed
 *                                   association: slabs run deepest first and hand ONE (nx,ny) plane of ϕtop per boundary to
 *                                   the slab above -- a grouped ncclSend / ncclRecv pair on a single-process communicator
 *                                   (ncclCommInitAll: RCCL over xGMI) between different devices; a device-to-device copy
 *                                   when the slabs share a device (every device id equal: tests on a one-GPU box);
 *   otmb_mgpu_transportmatrix_plan  every slab uploads its levels plus one halo level each side (neighbours only) from the
 *                                   host arrays -- nothing crosses devices; nnz = sums over the slabs (T: the union bound);
 *   otmb_mgpu_transportmatrix_fetch every slab fills its column range with global colptr offsets and copies it into its
 *                                   range of the caller's five matrices; T is compacted across slabs if entries cancelled.
 * device_ids: all different (RCCL; OTMB_MGPU_TRANSPORT=peer selects hipMemcpyPeerAsync) or all equal.  Errors: the status of
 * the first failing check in the reference's order over all slabs; otmb_mgpu_last_error names the slab.  An otmb_mgpu is not
 * shared between threads.  transport: 0 same-device copy, 1 RCCL, 2 peer copy.                                            */
typedef struct otmb_mgpu otmb_mgpu;
int32_t otmb_mgpu_create(int32_t ndev, const int32_t *device_ids, otmb_mgpu **out);
void otmb_mgpu_destroy(otmb_mgpu *mg);
const char *otmb_mgpu_last_error(const otmb_mgpu *mg);
int32_t otmb_mgpu_ndev(const otmb_mgpu *mg);
int32_t otmb_mgpu_transport(const otmb_mgpu *mg);
/* level bounds of the last facefluxes / plan: ndev + 1 entries, slab s = levels [bounds[s], bounds[s+1]) (0-based) */
int32_t otmb_mgpu_partition(const otmb_mgpu *mg, int64_t *bounds);
/* nslabs consecutive slabs of >= 1 level each with wet counts as even as a greedy sweep gets them (upper levels are wetter);
 * bounds: nslabs + 1 entries.  Pure host arithmetic, no GPU.                                                              */
int32_t otmb_balanced_partition(const int64_t *level_counts, int64_t nz, int32_t nslabs, int64_t *bounds);
/* The two promises of otmb_ctx_set_reuse_grid / otmb_ctx_set_reuse_fluxes, for the slabs (independent, both off by default): grid-constant
 * host arrays that are the very arrays of the previous plan (same pointers, same cut) are not uploaded again; ϕ that otmb_mgpu_facefluxes
 * computed and copied to these very host arrays is used where it is -- every slab keeps its levels on its device in the layout the plan
 * wants, with the one flux each halo level pushes into an owned cell filled in (ϕbottom above = the slab's first ϕtop,
 * src/velocities.jl:240; ϕtop below = the plane received from the slab below).  otmb_mgpu_uploaded_bytes: host -> device bytes so far. */
int32_t otmb_mgpu_set_reuse(otmb_mgpu *mg, int32_t reuse_grid, int32_t reuse_fluxes);
/* Speed only: the facefluxes chain hands its plane from slab to slab in `pieces` row bands (whole rows), so that slab s starts piece c
 * as soon as piece c of slab s + 1 has arrived: critical path of one field t_ff / W x (1 + (W - 1) / pieces) instead of t_ff
 * (src/velocities.jl:236-243).  0 (default): 4 on grids of 2^19 columns and more, else 1.  Same arrays for any number. */
int32_t otmb_mgpu_set_chain_pieces(otmb_mgpu *mg, int32_t pieces);
int64_t otmb_mgpu_uploaded_bytes(const otmb_mgpu *mg);
int32_t otmb_mgpu_facefluxes(otmb_mgpu *mg, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wet3d,
                             double fill, int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6]);
int32_t otmb_mgpu_transportmatrix_plan(otmb_mgpu *mg, const otmb_tm_args *args, int64_t nnz[5]);
int32_t otmb_mgpu_transportmatrix_fetch(otmb_mgpu *mg, int64_t *const colptr[5], int64_t *const rowval[5],
                                        double *const nzval[5], int64_t nnz_out[5]);
/* The same build in ONE call, pipelined over the slabs (speed only; the same five matrices bit for bit): the caller hands output arrays
 * of `capacity[m]` entries -- 7N, 7N, 5N, 3N, 3N always suffice (src/matrixbuilding.jl:244-296, :348-415, :450-477: a column of Tadv / T
 * holds at most the cell and its six neighbours, of TκH five, of TκVML / TκVdeep three) -- and gets the counts back in nnz_out; a host
 * binding wraps the first nnz_out[m] entries (Julia: unsafe_wrap with that length).  Slab s uploads its levels while slab s - 1 counts,
 * fills and copies its columns home, so the PCIe link carries both directions at once, which the two-phase protocol cannot do (every
 * upload precedes the count there, every download follows it): slabs that share a device take the link in turn for their uploads,
 * and a slab's column offsets are the running sums of the FINAL counts above it.  Meant for device_ids = {d, d, d, d}: four to eight
 * slabs on ONE device hide most of a time slice's upload behind its download; with one slab per device it saves the
 * nnz round trip.  OTMB_ERR_CAPACITY when an array is too small.                                                          */
int32_t otmb_mgpu_transportmatrix_onepass(otmb_mgpu *mg, const otmb_tm_args *args, int64_t *const colptr[5], int64_t *const rowval[5],
                                          double *const nzval[5], const int64_t capacity[5], int64_t nnz_out[5]);
/* Capacities that always suffice for otmb_mgpu_transportmatrix_onepass / otmb_transportmatrix_dev, from the wet mask ALONE (indices.wet3D
 * bytes; pure host arithmetic on all cores, no GPU, no context; once per grid): a column holds a row per wet neighbour of its cell (+ the
 * diagonal), whatever the fluxes are (src/velocities.jl:167-173 zeroes every flux towards land) -- so cap[OTMB_T] = cap[OTMB_TADV] = the union
 * pattern's count, cap[OTMB_TKH] and cap[OTMB_TKVDEEP] the horizontal / vertical neighbour counts (exact but for coinciding row-mates on
 * the tripolar seam), cap[OTMB_TKVML] = cap[OTMB_TKVDEEP].  About 24.6 N on an ocean grid against the 7 + 7 + 5 + 3 + 3 = 25 N of the per-column
 * maxima: what varies from time slice to time slice (Tadv: 3.9 N, TκVML: 0.6 N) is bounded by the host layers from the PREVIOUS slice's counts. */
int32_t otmb_static_capacity(const uint8_t *wet3d, int64_t nx, int64_t ny, int64_t nz, int32_t topology, int64_t cap[5]);

/* ---- makegridmetrics(; areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices) -- the array work of
 *      src/gridcellgeometry.jl:265-311 (device pointers; vertex permutation :158-178 and topology detection
 *      src/gridtopology.jl:33-53 are host decisions passed in as `perm` (0-based) and `topology`).
 * volcello (nx,ny,nz), areacello (nx,ny) with `missing` as NaN; fill_area/fill_vol: their _FillValue (NaN if none);
 * lon, lat (nx,ny); lon_vertices, lat_vertices (4,nx,ny) as given.  Outputs: area2d, v3d, thkcello, z3d and the
 * three groups of four (nx,ny) arrays in OTMB_DIR_* order: edge_length_2D, distance_to_edge_2D,
 * distance_to_neighbour_2D.  Transcendentals use the device math library: ~1e-15 relative to a host build. */
int32_t otmb_makegridmetrics_dev(otmb_ctx *ctx, const double *volcello, const double *areacello, double fill_area,
                                 double fill_vol, const double *lon, const double *lat, const double *lon_vertices,
                                 const double *lat_vertices, const int32_t perm[4], int64_t nx, int64_t ny, int64_t nz,
                                 int32_t topology, double *area2d, double *v3d, double *thkcello, double *z3d,
                                 double *const edge_length[4], double *const dist_edge[4], double *const dist_nbr[4]);
/* the same on host arrays (what a Julia / C caller has): inputs up, the seventeen derived arrays back; blocking */
int32_t otmb_makegridmetrics(otmb_ctx *ctx, const double *volcello, const double *areacello, double fill_area, double fill_vol,
                             const double *lon, const double *lat, const double *lon_vertices, const double *lat_vertices,
                             const int32_t perm[4], int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *area2d, double *v3d,
                             double *thkcello, double *z3d, double *const edge_length[4], double *const dist_edge[4],
                             double *const dist_nbr[4]);

/* ---- bolus_GM_velocity(ρ, gridmetrics, indices; κGM = 600, maxslope = 0.01) -- src/RediGM.jl:46-79 with
 *      globalverticalfacetriadderivative (src/triads.jl:84-146) and globalverticaldyadderivative
 *      (src/dyads.jl:38-78).  Experimental in the reference, not connected to transportmatrix, and unpinned by
 *      any reference test.  rho, z3d (gridmetrics.Z3D): (nx,ny,nz); wet3d: indices.wet3D bytes; dist_east /
 *      dist_north: gridmetrics.distance_to_neighbour_2D[:east] / [:north].  u, v: (nx,ny,nz), NaN off Lwet.
 *      A bipolar topology is an error (the reference evaluates k₋₁(nothing) at j == ny).               */
int32_t otmb_bolus_gm_velocity_dev(otmb_ctx *ctx, const double *rho, const double *z3d, const uint8_t *wet3d,
                                   const double *dist_east, const double *dist_north, int64_t nx, int64_t ny,
                                   int64_t nz, int32_t topology, double kappa_gm, double maxslope, double *u, double *v);
int32_t otmb_bolus_gm_velocity(otmb_ctx *ctx, const double *rho, const double *z3d, const uint8_t *wet3d,
                               const double *dist_east, const double *dist_north, int64_t nx, int64_t ny, int64_t nz,
                               int32_t topology, double kappa_gm, double maxslope, double *u, double *v);

/* ---- the reference's two-step formulation as a general (non-fused) device path -------------------------
 * otmb_sparse_entries_*: the COO generators in the reference's emission order -- which = 0
 *   advection_operator_sparse_entries (src/matrixbuilding.jl:221-299), 1 horizontal_diffusion_… (:337-418),
 *   2 vertical_diffusion_… with the mixed-layer mask (:438-479, Ω of :85), 3 the same with Ω = all (:109).
 *   plan -> number of triplets (and the reference's errors), fill -> I, J, V (device arrays of that length).
 * otmb_sparse_*: SparseArrays.sparse(I, J, V, m, n) as called at :41,63,92,116 -- duplicates summed in input
 *   order, explicit zeros kept, rows ascending per column.  plan -> nnz, fill -> colptr (n+1), rowval, nzval.
 * Device pointers; one plan/fill pair at a time per context.                                             */
int32_t otmb_sparse_entries_plan_dev(otmb_ctx *ctx, int32_t which, const otmb_tm_args *args, int64_t *len);
int32_t otmb_sparse_entries_fill_dev(otmb_ctx *ctx, int64_t *I, int64_t *J, double *V);
int32_t otmb_sparse_plan_dev(otmb_ctx *ctx, const int64_t *I, const int64_t *J, const double *V, int64_t len, int64_t m,
                             int64_t n, int64_t *nnz);
int32_t otmb_sparse_fill_dev(otmb_ctx *ctx, int64_t *colptr, int64_t *rowval, double *nzval);

/* ---- A + B for SparseMatrixCSC{Float64,Int64} -- SparseArrays' map(+, A, B), the `+` of
 *      src/matrixbuilding.jl:147 used when the caller passes precomputed operators (:133-143): per column a
 *      sorted merge, a missing operand counts as +0.0, results that are exactly zero are not stored.
 * _dev: device pointers, two-phase (plan -> nnz, fill).  otmb_spadd: host pointers, Ci/Cx capacity nnz(A)+nnz(B). */
int32_t otmb_spadd_plan_dev(otmb_ctx *ctx, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                            const int64_t *Bp, const int64_t *Bi, const double *Bx, int64_t *nnz_out);
int32_t otmb_spadd_fill_dev(otmb_ctx *ctx, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                            const int64_t *Bp, const int64_t *Bi, const double *Bx, int64_t *Cp, int64_t *Ci, double *Cx);
int32_t otmb_spadd(otmb_ctx *ctx, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, const int64_t *Bp,
                   const int64_t *Bi, const double *Bx, int64_t *Cp, int64_t *Ci, double *Cx, int64_t *nnz_out);

/* ---- lump_and_spray(wet3D, vol, T, mask = trues(size(wet3D)); di = 2, dj = 2, dk = 1) -- src/extratools.jl:38-119:
 *      the coarsening operators LUMP (Nc x N, volume weighted: LUMP * x is the coarse vector), SPRAY (N x Nc, ones:
 *      LUMP * T * SPRAY is the coarse operator) and the coarse volumes vol_c.  Cells are lumped in di x dj x dk blocks,
 *      inside `mask` only, never across cells that T's pattern does not connect (connected components of the block,
 *      Graphs.connected_components; T's pattern must be symmetric inside every block like Graphs.SimpleGraph demands:
 *      OTMB_ERR_ASYMMETRIC_PATTERN otherwise -- e.g. an advection-only operator).  Only the PATTERN of T is read.
 * _dev (device pointers, two-phase): plan takes wet3d / lwet3d / lwet / n_wet as makeindices returns them, mask (bytes,
 *   NULL = all true) and T's colptr / rowval; it returns Nc.  fill writes LUMP (colptr N+1 = 1..N+1: one entry per column,
 *   rowval N, nzval N), SPRAY (colptr Nc+1, rowval N, nzval N) and vol_c (Nc).  Limits: nx <= 4900, di*dj*dk <= 4096.
 * otmb_lump_and_spray (host pointers, what the Julia shim calls): outputs with capacity N (N+1 for spray_colptr); the
 *   trivial arrays (LUMP's colptr, SPRAY's ones) are left to the caller.                                            */
int32_t otmb_lump_and_spray_plan_dev(otmb_ctx *ctx, const uint8_t *wet3d, const uint8_t *mask, const int64_t *lwet3d,
                                     const int64_t *lwet, int64_t n_wet, int64_t nx, int64_t ny, int64_t nz,
                                     const int64_t *t_colptr, const int64_t *t_rowval, int64_t di, int64_t dj, int64_t dk,
                                     int64_t *n_coarse);
int32_t otmb_lump_and_spray_fill_dev(otmb_ctx *ctx, const double *vol, int64_t *lump_colptr, int64_t *lump_rowval,
                                     double *lump_nzval, int64_t *spray_colptr, int64_t *spray_rowval, double *spray_nzval,
                                     double *vol_c);
int32_t otmb_lump_and_spray(otmb_ctx *ctx, const uint8_t *wet3d, const uint8_t *mask, int64_t nx, int64_t ny, int64_t nz,
                            const double *vol, int64_t n_wet, const int64_t *t_colptr, const int64_t *t_rowval, int64_t di,
                            int64_t dj, int64_t dk, int64_t *lump_rowval, double *lump_nzval, int64_t *spray_colptr,
                            int64_t *spray_rowval, double *vol_c, int64_t *n_coarse);

#ifdef __cplusplus
}
#endif
#endif /* OTMB_H */
