/*
 * mirt.h -- C-ABI of the MI355X-native per-pixel render path ("mirt").
 *
 * This is the drop-in boundary for the two hot loops of ArchDD/CPP-Raytracer-Rasterizer.  The reference
 * has no plugin / FFI layer: its hot path is the free function `void Draw()` reading globals and writing
 * through `PutPixelSDL` (raytracer/Source/raytracer.cpp:104,547; rasteriser/Source/rasteriser.cpp:86,461;
 * raytracer/Source/SDLauxiliary.h:29,70-81).  Each entry point below names the reference interface it
 * replaces.  The host-side `Draw()` adapter that marshals the reference-shaped globals into these calls is
 * cpp-raytracer-rasterizer_amd/host/mirt_draw.hpp; INTEGRATION.md shows the lines a maintainer adds.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns 0 on success or a negative
 *     mirt_status, with a human-readable message available from mirt_last_error();
 *   - the library never exits the process and never falls back to a CPU path: without a usable GPU
 *     every compute entry point fails with MIRT_ERR_NO_DEVICE;
 *   - the caller owns every host pointer for the duration of a call; the library owns all device memory;
 *   - entry points are meant for one host thread (the SDL loop), are not re-entrant, and are synchronous
 *     (the frame is complete on return) unless the name ends in `_device`/`_async`;
 *   - a triangle is 15 floats {v0, v1, v2, normal, color} -- the field order of `class Triangle`
 *     (raytracer/Source/TestModel.h:11-32; sizeof == 60); a light is 7 floats {position, color,
 *     intensity} == `class Light` (TestModel.h:35-45; sizeof == 28); matrices are GLM column-major mat3;
 *   - per-pixel planes are row-major with row stride == width (the reference indexes them with
 *     SCREEN_HEIGHT, which is only correct for its square 500x500 frame -- SURVEY Appendix E-1).
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_API __attribute__((visibility("default")))
#define MIRT_ABI_VERSION 4
#define MIRT_MAX_LIGHTS 32            /* Light lights[32], raytracer.cpp:48 / rasteriser.cpp:50 */

typedef enum mirt_status {
    MIRT_OK = 0,
    MIRT_ERR_NO_DEVICE = -1,          /* no HIP device / wrong architecture                     */
    MIRT_ERR_NOT_INITIALISED = -2,    /* mirt_init not called                                    */
    MIRT_ERR_INVALID_ARGUMENT = -3,
    MIRT_ERR_NO_SCENE = -4,           /* render call before mirt_scene_upload                    */
    MIRT_ERR_OUT_OF_MEMORY = -5,
    MIRT_ERR_HIP = -6                 /* a HIP runtime call failed; see mirt_last_error()        */
} mirt_status;

/* cameraPos / cameraRot / focalLength / SCREEN_WIDTH / SCREEN_HEIGHT (raytracer.cpp:64-73,
 * rasteriser.cpp:35-41). rot is GLM column-major: rot[c*3+r] == cameraRot[c][r]. */
typedef struct mirt_view {
    float pos[3];
    float rot[9];
    float focal;
    int32_t width;
    int32_t height;
} mirt_view;

/* class Light (TestModel.h:35-45), same field order and size (28 bytes). */
typedef struct mirt_light {
    float pos[3];
    float color[3];
    float intensity;
} mirt_light;

/* How the ray tracer finds each ray's candidate triangles.  All modes return identical results. */
typedef enum mirt_rt_mode {
    MIRT_RT_AUTO = 0,     /* brute force for small scenes, binned otherwise                          */
    MIRT_RT_BRUTE = 1,    /* every ray tests every triangle (what ClosestIntersection does)         */
    MIRT_RT_BINNED = 2    /* conservative screen-tile / light-cube binning, same accept arithmetic  */
} mirt_rt_mode;

/* Counters of the last render call (ray accounting follows SURVEY section 8(d)). */
typedef struct mirt_stats {
    uint64_t primary_rays;        /* width * rows rendered                                          */
    uint64_t shadow_rays;         /* nlights * (pixels whose primary ray hit)                       */
    uint64_t tests;               /* ray-triangle tests actually executed (0 if not counted: the binned kernel keeps this and
                                     the four counts below for frames rendered with mirt_set_profiling(1) only)          */
    float gpu_ms;                 /* hipEvent time of the call's device work; 0 unless profiling is on   */
    float kernel_ms[8];           /* per-kernel time of the call when profiling is on (see below)    */
    int32_t mode_used;            /* mirt_rt_mode actually used                                      */
    uint64_t candidates;          /* ray-candidate pairs the frame's lists offered (>= tests: the binned kernel skips
                                     candidates that cannot matter before it tests them); == tests elsewhere */
    /* the binned ray tracer's own loop counts (0 elsewhere): wave-level steps of the two filter loops -- one step = one
     * candidate row tested by the lanes of a wave -- and exact-stage drains; what the kernel's instruction count scales with */
    uint64_t steps_primary;
    uint64_t steps_shadow;
    uint64_t drains;
    /* binned ray tracer (ABI 4): 1 when the frame kept the CAMERA's binning pass of its stream -- nothing that pass depends on had
     * changed since (the view stood still: a light key, a toggle; raytracer.cpp:385-537) --, and how many of the scene's triangles
     * the rows of the call can see at all (what the camera's pass walked; 0 when it was kept) */
    uint32_t bins_reused;
    uint32_t selected_triangles;
} mirt_stats;

/* indices into mirt_stats.kernel_ms */
enum { MIRT_K_PREP = 0, MIRT_K_BIN = 1, MIRT_K_TRACE = 2, MIRT_K_DOF = 3,
       MIRT_K_RASTER_SETUP = 4, MIRT_K_RASTER_FRAG = 5, MIRT_K_RASTER_RESOLVE = 6, MIRT_K_CLEAR = 7 };

/* ---- lifetime ------------------------------------------------------------------------------------ */

/* One context per process, driven from one thread at a time (the reference is a single-threaded main loop); the calls
 * are not re-entrant.  Several processes may share a GPU (one process per GPU under torch.distributed). */

/* Replaces nothing in the reference (it has no device).  Selects HIP device `device` (use the process's
 * LOCAL_RANK under torch.distributed), checks it is gfx950, creates the library stream. */
MIRT_API int mirt_init(int device);
MIRT_API void mirt_shutdown(void);
MIRT_API const char *mirt_last_error(void);
MIRT_API int mirt_abi_version(void);
/* When on, every call and every kernel launch is bracketed by hipEvents on the library stream and
 * mirt_stats.gpu_ms / kernel_ms are filled (costs a few microseconds per event; off by default). */
MIRT_API int mirt_set_profiling(int on);
/* Blocks until all work queued by the library has finished. */
MIRT_API int mirt_sync(void);
/* The library's hipStream_t (as void*), so a caller can order its own work around ours (wait for an event before a
 * call, record one after it).  With several frames in flight (below) it is the stream of the most recent call only:
 * order with mirt_sync() instead. */
MIRT_API void *mirt_stream(void);
/* How many *_device frames may be in flight at once: 1 (default; calls run in order on one stream) up to 4 (calls take
 * that many streams in turn; the next frames are dispatched, and run where the device has room, while the previous ones
 * drain -- the kernels of a frame are short dependent chains, and several frames fill each other's gaps).  With k frames
 * in flight the caller must give k consecutive frames DIFFERENT output planes -- the double (k-fold) buffering a render
 * loop that presents one frame while drawing the next already has (the reference's SDL_UpdateRect after Draw(),
 * raytracer.cpp:653); frames i and i + k share a stream and may share planes -- and mirt_sync() before reading them.
 * Every scratch buffer a frame writes (origin and bin tables, raster keys, depth-of-field planes, counters) exists once
 * per stream. */
MIRT_API int mirt_set_frames_in_flight(int frames);

/* ---- host surfaces --------------------------------------------------------------------------------- */

/* The surface PutPixelSDL writes (SDL_Surface::pixels of InitializeSDL's SDL_SetVideoMode, SDLauxiliary.h:31-52) is ordinary
 * pageable memory; a frame copied into it is staged by the runtime at about half the link rate.  Registering it once --
 * pixels, h * pitch bytes -- pins and maps it: mirt_raytrace / mirt_rasterise / their _ex forms then deliver every frame whose
 * out_xrgb rows lie inside a registered surface by one DMA copy into the pinned pages (MIRT_HOST_PATH=direct: the render
 * kernels store the XRGB words straight into the mapped surface instead, no staging plane).  On the MI355X box of this
 * repository's measurements neither beats the pageable path (the link gives 37-46 GB/s either way, see bench.py host_path);
 * the calls exist for hosts whose pageable copies are slower.  The caller keeps the memory alive until
 * mirt_surface_unregister / mirt_shutdown.  Up to 4 surfaces. */
MIRT_API int mirt_surface_register(void *pixels, size_t bytes);
MIRT_API int mirt_surface_unregister(void *pixels);

/* Asynchronous frames into a registered surface: the frame is rendered and copied (one stream-ordered DMA copy into the
 * pinned pages) on the library stream and the call returns as soon as both are queued; mirt_sync() -- the SDL_UpdateRect of the
 * loop (raytracer.cpp:653, rasteriser.cpp:528) -- completes it.  With mirt_set_frames_in_flight(2) the copy of frame i overlaps
 * the render of frame i + 1, which is what makes the host boundary run at the rate of the link (bench.py host_path.async):
 * consecutive frames then need DIFFERENT surfaces (double buffering), exactly as the *_device calls do.  out_xrgb must lie in a
 * surface registered above (MIRT_ERR_INVALID_ARGUMENT otherwise: pageable memory cannot take a stream-ordered copy).  Same
 * pixels as mirt_raytrace / mirt_rasterise: interior only for the ray tracer, the whole surface for the rasteriser. */
MIRT_API int mirt_raytrace_async(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                 int mode, uint32_t *out_xrgb, int pitch_bytes);
MIRT_API int mirt_rasterise_async(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                  uint32_t *out_xrgb, int pitch_bytes);

/* ---- scene --------------------------------------------------------------------------------------- */

/* Replaces the global `vector<Triangle> triangles` (raytracer.cpp:28, rasteriser.cpp:64): copies n
 * triangles (n x 15 floats, reference AoS order) to the device and precomputes the per-triangle edge
 * data.  `culled` (nullable, n bytes) is Triangle::isCulled of the rasteriser (rasteriser/Source/
 * TestModel.h:18); the ray tracer ignores it.  The caller may free both arrays on return. */
MIRT_API int mirt_scene_upload(const float *tris15, const uint8_t *culled, int n);
/* Replaces the per-frame `triangles[i].isCulled = ...` writes of the rasteriser's Update()
 * (rasteriser.cpp:404-447) for an already uploaded scene. */
MIRT_API int mirt_scene_set_culled(const uint8_t *culled, int n);
MIRT_API int mirt_scene_size(void);

/* LoadTestModel (raytracer/Source/TestModel.h:51-192): writes the 30 Cornell-box triangles, returns 30. */
MIRT_API int mirt_scene_cornell(float *tris15);
/* Synthetic soup for the benchmark configs (no reference counterpart; generator defined in SURVEY
 * section 8(d)): mt19937(seed), centres U[-1,1]^3, two edges U[-s,s]^3, colours U[0.15,0.75]^3,
 * normal by the reference's Triangle::ComputeNormal rule. */
MIRT_API int mirt_scene_soup(uint32_t seed, int n, float s, float *tris15);
/* The cull step of the rasteriser's Update() (rasteriser.cpp:385-447, InCuboid :451-458), host side.
 * flags: bit0 = BACKFACE_CULLING_ENABLED, bit1 = FRUSTUM_CULLING_ENABLED (both default on, :25-26). */
MIRT_API int mirt_cull(const float *tris15, int n, const mirt_view *view, int flags, uint8_t *culled);
/* The same step on the device for the uploaded scene (one thread per triangle; the flags go straight into the scene's
 * device-side cull array, no host copy either way) -- what replaces the loop at rasteriser.cpp:404-447 once meshes are
 * large.  mirt_scene_get_culled reads the flags back (the reference's triangles[i].isCulled).  The flags are those of the
 * NEXT mirt_rasterise* call(s), as Update() culls right before Draw(): with several frames in flight they are written for,
 * and on, the stream the next call takes, so the frames still running on the other streams keep their own; a rasterised
 * frame that lands on another stream (the second row band of a frame, or after a ray-traced frame took that turn) brings
 * the most recent flags over, ordered behind the cull kernel. */
MIRT_API int mirt_cull_device(const mirt_view *view, int flags);
MIRT_API int mirt_scene_get_culled(uint8_t *culled, int n);
/* LoadSTL::LoadSTLFile (rasteriser/Source/LoadSTL.cpp:17-97): reads an ASCII STL the way the reference does (every line
 * containing "outer" is followed by three vertex lines), multiplies every coordinate by -scale (the reference: 0.05f),
 * sets the colour (the reference: 0.5, 0.5, 0.5) and recomputes the normals.  Returns the number of facets in the file;
 * writes the first max_tris of them when tris15 is not NULL (call with NULL first to size the array). */
MIRT_API int mirt_scene_load_stl(const char *path, float scale, const float *colour3, float *tris15, int max_tris);

/* Soft shadows (SOFT_SHADOWS_ENABLED / SOFT_SHADOWS_SAMPLES / randomPositions, raytracer.cpp:40-41,84,186-190,
 * 272-287): when samples > 1 every light k is replaced in DirectLight by `samples` jittered positions
 * positions[(k*samples + i)*3 .. +2] with 1/samples of its power each.  The caller generates the positions (the
 * reference draws them with rand() in AddLight; host/mirt_draw.hpp does the same).  samples <= 1 switches back to
 * hard shadows.  lights x samples may not exceed MIRT_MAX_LIGHTS. */
MIRT_API int mirt_set_soft_shadows(int samples, const float *positions, int npositions);

/* Depth of field (DOF_ENABLED / DOF_KERNEL_SIZE / FOCAL_LENGTH, raytracer.cpp:43-45,613-640; rasteriser.cpp:29-31,
 * 494-513): kernel_size > 1 blurs pixelColours with a kernel_size x kernel_size stencil weighted by the centre pixel's
 * |distance - focal_length| before PutPixelSDL, for both renderers (the reference uses 8 with 1.3 / 1.9).  Taps whose
 * flat index leaves the frame are undefined behaviour in the reference; here they contribute nothing.  out_rgb keeps
 * the unblurred pixelColours, as in the reference.  kernel_size <= 1 switches it off. */
MIRT_API int mirt_set_depth_of_field(int kernel_size, float focal_length);

/* Supersampling (AA_ENABLED / AA_SAMPLES, raytracer.cpp:37-38,549-599): samples > 1 fires samples x samples sub-rays
 * per pixel and averages them, reproducing the reference's loop exactly (the pixel's closest-hit record is carried
 * across the sub-rays, x1 only advances after a sub-ray that hit).  samples <= 1 switches it off.  The reference uses 3. */
MIRT_API int mirt_set_antialiasing(int samples);

/* ---- ray tracer: replaces Draw() + CalculateDOF() of raytracer.cpp:547-656 -------------------------- */

/* One frame into host buffers.  out_xrgb (required) is the SDL surface's `pixels` (XRGB8888, the words
 * PutPixelSDL stores, SDLauxiliary.h:75-80) with `pitch_bytes` per row; as in the reference only interior
 * pixels x in [1,W-2], y in [1,H-2] are written (raytracer.cpp:618-620) -- border words are left untouched.
 * out_rgb (nullable, W*H*3 floats) receives `pixelColours` (raytracer.cpp:88,600); out_index (nullable,
 * W*H int32) receives closestIntersections[].triangleIndex, -1 where the primary ray missed (:98,243-247). */
MIRT_API int mirt_raytrace(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                           int mode, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb, int32_t *out_index);

/* Same frame, rows [y0, y1) only, into DEVICE buffers the caller owns (e.g. torch tensors), queued on the
 * library stream without a host sync; used for screen-band sharding across GPUs and for benchmarking with
 * resident buffers.  d_xrgb points at row 0 of a full-frame (or band-relative, see row_origin) surface:
 * row y is written at d_xrgb + (y - row_origin) * pitch_bytes.  d_rgb / d_index (nullable) likewise use
 * row stride W.  Interior-only rule as above (relative to the FULL frame). */
MIRT_API int mirt_raytrace_device(const mirt_view *view, const mirt_light *lights, int nlights,
                                  const float *indirect, int mode, int y0, int y1, int row_origin,
                                  void *d_xrgb, int pitch_bytes, void *d_rgb, void *d_index);

/* The same two calls with the rest of `struct Intersection` (raytracer.cpp:91-98): out_distance / d_distance (nullable, W*H
 * floats, row stride W) receives closestIntersections[].distance -- the FLT_MAX of Update()'s reset (:335-339) where the
 * primary ray missed -- and out_position / d_position (nullable, W*H*3 floats) closestIntersections[].position (0 where it
 * missed; the reference leaves it uninitialised).  With depth of field on, the rows the blur reads beyond the band are
 * rendered too and their entries of these two planes are written as well. */
MIRT_API int mirt_raytrace_ex(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                              int mode, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb, int32_t *out_index,
                              float *out_distance, float *out_position);
MIRT_API int mirt_raytrace_device_ex(const mirt_view *view, const mirt_light *lights, int nlights,
                                     const float *indirect, int mode, int y0, int y1, int row_origin,
                                     void *d_xrgb, int pitch_bytes, void *d_rgb, void *d_index,
                                     void *d_distance, void *d_position);

/* ---- rasteriser: replaces Update()'s clear + Draw() + CalculateDOF() of rasteriser.cpp:183-192,461-529 */

/* One frame into host buffers.  Every word of out_xrgb is written: the whole surface is cleared to black
 * (rasteriser.cpp:190) and interior pixels then receive the resolved colour (:491-519).  out_rgb (nullable)
 * = pixelColours, out_zinv (nullable) = depthBuffer (1/z, 0 where nothing was drawn, :52,188), out_index
 * (nullable) = index of the triangle that owns each pixel or -1. */
MIRT_API int mirt_rasterise(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                            uint32_t *out_xrgb, int pitch_bytes, float *out_rgb, float *out_zinv,
                            int32_t *out_index);

/* Rows [y0, y1) into DEVICE buffers, queued on the library stream (see mirt_raytrace_device). */
MIRT_API int mirt_rasterise_device(const mirt_view *view, const mirt_light *lights, int nlights,
                                   const float *indirect, int y0, int y1, int row_origin,
                                   void *d_xrgb, int pitch_bytes, void *d_rgb, void *d_zinv, void *d_index);

/* ---- several GPUs of one node: frames shard by row bands ----------------------------------------- */

/* One process per GPU, as under torch.distributed / MPI; the reference itself is a single process (its only parallelism is
 * OpenMP over rows, raytracer.cpp:557): these calls are what a sharded host loop adds around Draw().  Every rank uploads the
 * same scene (the triangle list is replicated) and makes the same calls in the same order.
 *
 * mirt_band_of          rows [y0, y1) of `rank` when `height` rows are split into `world` contiguous bands (sizes differ by at
 *                       most one row).  Pure arithmetic, no device needed.
 * mirt_comm_create_id   rank 0: creates the 128-byte id of a new group (ncclGetUniqueId); the caller hands it to the other
 *                       ranks by its own means (MPI_Bcast, torch.distributed.broadcast_object_list, a file).
 * mirt_comm_init        every rank, after mirt_init: joins the group (collective).  Transport: RCCL point-to-point over xGMI,
 *                       loaded at run time; MIRT_COMM=shm selects a host-staged loopback for ranks that share one GPU (tests).
 * mirt_*_sharded        renders THIS rank's band of `nviews` consecutive frames (views[0..nviews) share a frame size) and
 *                       gathers the XRGB bands on rank `root` -- the one exchange step of the path: each peer sends its rows
 *                       straight to the root on its own link (a direct gather, not a ring), grouped into one collective per
 *                       call, so frames that render in microseconds can travel many to a gather.  d_frames (device memory,
 *                       required on the root only): nviews frames of height * pitch_bytes, pitch_bytes == 4 * width; frame i
 *                       of the call is byte-identical to a single-GPU mirt_*_device frame (border words of received bands,
 *                       which the ray tracer never writes, arrive as 0).  Asynchronous: the gather runs on a communication
 *                       stream and overlaps the next call's render (two band buffers per rank); mirt_sync() waits for both.
 *                       Without mirt_comm_init (or world == 1) the calls simply render whole frames.  Needs
 *                       mirt_set_frames_in_flight(1). */
#define MIRT_COMM_ID_BYTES 128
MIRT_API int mirt_band_of(int rank, int world, int height, int *y0, int *y1);
/* The messages of one gather (what mirt_*_sharded sends): piece i moves bytes[i] from band_offset[i] of rank peer[i]'s band
 * buffer (nviews bands of its rows, 4 * width bytes per row) to root_offset[i] of the root's frame buffer.  Returns the number
 * of pieces (writes at most max_pieces; arrays nullable).  Pure arithmetic, no device needed. */
MIRT_API int mirt_band_plan(int world, int root, int width, int height, int nviews, uint64_t *root_offset, uint64_t *band_offset,
                            uint64_t *bytes, int32_t *peer, int max_pieces);
/* The partition of a sharded frame among the ranks (every rank must set the same; default 0).  The setting belongs to the library's
 * context: mirt_shutdown() puts it back to 0, so set it after mirt_init().  Nothing exchanges or verifies it: ranks with different strip
 * heights build different gather plans and the grouped send / receive then mismatches (RCCL waits instead of failing) -- the weighted
 * partition has no per-rank parameter to disagree about and is what bench.py uses for binned frames.
 *
 *   strip_rows == 0   contiguous bands (mirt_band_of): one binning pass and one launch chain per rank and frame -- the right
 *                     choice for the binned ray tracer, whose per-frame cost has a part that does not shrink with the rows;
 *   strip_rows  > 0   interleaved strips of that many rows (a multiple of 8), strip s to rank s % world: every rank samples the
 *                     whole height of the frame, which evens out scenes whose cost is concentrated in some rows -- what the
 *                     reference's `#pragma omp parallel for schedule(auto)` over rows does (raytracer.cpp:557, rasteriser.cpp:467),
 *                     at the granularity a GPU launch needs (64).  Each strip is a launch chain of its own.
 * mirt_partition_segments / mirt_partition_plan: a rank's row segments [y0[k], y1[k]) and the gather's messages for either
 * partition (mirt_band_of / mirt_band_plan are the strip_rows == 0 case); a band buffer holds a rank's segments of one view back
 * to back.  Pure arithmetic, no device needed; both return the count (arrays nullable, at most max_* entries written).
 *   MIRT_PARTITION_WEIGHTED   contiguous bands of equal ESTIMATED COST instead of equal height: the counterpart of
 *                     `schedule(auto)` for a frame whose rows differ in what they cost (a triangle soup seen in perspective: the
 *                     middle bands of BASELINE configs[4] hold twice the candidates of the outer ones).  The first kernel of
 *                     a binned ray-traced frame leaves an estimate of the (tile, triangle) pairs per tile row of the WHOLE frame;
 *                     every rank computes the same numbers from the same scene and view, and the bands of sharded call c come
 *                     from the histogram of call c - 2 by integer arithmetic -- identical on every rank, nothing exchanged.
 *                     Equal bands until a histogram exists (the first two calls, frames that take another path).
 * mirt_weighted_bounds: that arithmetic as a pure function (world + 1 boundaries, multiples of 8 rows, from a histogram of
 * hist_rows coarse tile rows of (1 << hist_shift) tile rows each); mirt_partition_bounds: the boundaries the NEXT sharded call
 * will use; mirt_bounds_plan: the gather's messages for explicit boundaries; mirt_set_cost_histogram(1) makes binned frames
 * leave the histogram outside sharded calls too and mirt_cost_histogram returns the latest one (returns its row count, 0: none). */
#define MIRT_PARTITION_WEIGHTED (-1)
MIRT_API int mirt_set_partition(int strip_rows);
MIRT_API int mirt_set_cost_histogram(int on);
MIRT_API int mirt_cost_histogram(uint32_t *hist, int max_rows, int *rows, int *shift);
MIRT_API int mirt_weighted_bounds(const uint32_t *hist, int hist_rows, int hist_shift, int width, int height, int world, int32_t *bounds);
MIRT_API int mirt_partition_bounds(int world, int width, int height, int32_t *bounds);
MIRT_API int mirt_bounds_plan(int world, int root, int width, int height, int nviews, const int32_t *bounds, uint64_t *root_offset,
                              uint64_t *band_offset, uint64_t *bytes, int32_t *peer, int max_pieces);
MIRT_API int mirt_partition_segments(int rank, int world, int height, int strip_rows, int32_t *y0, int32_t *y1, int max_segments);
MIRT_API int mirt_partition_plan(int world, int root, int width, int height, int nviews, int strip_rows, uint64_t *root_offset,
                                 uint64_t *band_offset, uint64_t *bytes, int32_t *peer, int max_pieces);
MIRT_API int mirt_comm_create_id(void *id128);
MIRT_API int mirt_comm_init(const void *id128, int rank, int world);
MIRT_API int mirt_comm_shutdown(void);
/* Link check after mirt_comm_init: `bytes` of a pattern travel from this rank to itself through the group's transport (one send
 * and one receive in a group, exactly as in a gather) and are compared.  Any world size; synchronous. */
MIRT_API int mirt_comm_selfcheck(size_t bytes);
MIRT_API int mirt_raytrace_sharded(const mirt_view *views, int nviews, const mirt_light *lights, int nlights, const float *indirect,
                                   int mode, int root, void *d_frames, int pitch_bytes);
MIRT_API int mirt_rasterise_sharded(const mirt_view *views, int nviews, const mirt_light *lights, int nlights, const float *indirect,
                                    int root, void *d_frames, int pitch_bytes);

/* Counters / timings of the most recent render call. */
MIRT_API int mirt_get_stats(mirt_stats *out);
/* With profiling on and two or more frames in flight: the per-kernel GPU times (kernel_ms8[8], indexed like mirt_stats.kernel_ms) and the
 * GPU time of the call BEFORE the last one -- the frame on the other stream, whose kernels ran while the frames on both sides of
 * it were running, which is the state a render loop is in.  (mirt_get_stats reports the last call, whose tail runs alone.)
 * Waits for all streams. */
MIRT_API int mirt_get_previous_kernel_ms(float *kernel_ms8, float *gpu_ms);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
