// mirt_capi.hip -- the C-ABI of include/mirt.h on top of the HIP kernels.  Owns the device scene, the
// library stream, staging buffers and the per-call statistics.  No CPU fallback: without a gfx950 device
// every compute entry point returns MIRT_ERR_NO_DEVICE.
#include "bin_sort.hpp"
#include "comm.hpp"
#include "cull.hpp"
#include "dof.hpp"
#include "rt_common.hpp"
#include "raster_common.hpp"
#include "rt_binned.hpp"
#include "scan.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <vector>

namespace mirt {

// kernels (rt_kernels.hip, raster_kernels.hip)
__global__ void k_prep_origin(const float *, int, const float *, v3, int, OriginRow *, OriginRow *, uint32_t *, unsigned long long *, uint32_t *);
template <int P> __global__ void k_rt_brute(const RtFrame);
template <int P> __global__ void k_rt_small(const RtFrame, int);
__global__ void k_rt_wave(const RtFrame);
struct RtTileFrame {
    RtFrame f;
    BinFrameDesc cam;
    int tiles_x, tiles_y;
    unsigned long long *clear_hits;
    float4 *tables;
};
__global__ void k_tile_tables(const RtTileFrame);
template <int TW, bool AA> __global__ void k_rt_tile2(const RtTileFrame);
template <int WG> __global__ void k_bin_pairs(const float *, const OriginRow *, const OriginRow *, int, BinSet, BinPairs);
struct TilePairRec { uint32_t tile, beg, nA, nB; };
constexpr int ORDER_CLASSES = 8, ORDER_GROUPS = 8;
struct RtTraceFrame {                            // (rt_trace.hip)
    RtFrame f;
    const uint32_t *cam_off;
    const uint32_t *cam_entries;
    const GeoRow *geo;
    const ShadeRow *shade;
    const uint32_t *light_off;
    const LightRow *light_rows;
    const uint32_t *light_tri;
    const BinFrameDesc *light_frames;
    int tiles_x;
    int cube_bins;
    int cam_shells;
    int light_shells;
    const uint32_t *pair_count;
    uint32_t pair_cap;
    const TilePairRec *order;
    const uint32_t *order_count;
    uint32_t order_seg;
    const uint32_t *sel;
    const uint32_t *sel_count;
    int lazy_geo;
    const uint32_t *light_pair_count;
    uint32_t light_pair_cap;
};
template <bool AA, bool STATS, int WAVES = 4> __global__ void k_rt_trace2(const RtTraceFrame);
__global__ void k_prep_select(const float *, int, const BinFrameDesc, const SelectOut);
__global__ void k_select_faces(const float *, int, const float *, const BinFrameDesc *, OriginRow *, uint32_t *, uint32_t, uint32_t *);
__global__ void k_tile_order(const uint32_t *, int, int, int, int, uint32_t *, uint32_t, TilePairRec *, uint32_t);
__global__ void k_geo_table(const float *, int, GeoRow *, ShadeRow *);
__global__ void k_expand_light_rows(const uint32_t *, const uint32_t *, int, uint32_t, const OriginRow *, int, LightRow *, const uint32_t *, uint32_t, uint32_t *);
size_t rt_trace_lds_bytes(int waves);
int launch_raster(RasterFrame &f, RasterScratch &s, uint64_t scene_version, hipStream_t stream, hipEvent_t *ev, bool *ev_used);
__global__ void k_cull(const float *, int, const CullParams, uint8_t *);

namespace {

char g_err[512] = "no error";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(MIRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum { EV_CALL0 = 0, EV_CALL1 = 1, EV_K0 = 2, EV_COUNT = 2 + 2 * 8 };
constexpr int MAX_FLIGHT = 4;                    // most frames in flight (mirt_set_frames_in_flight): one HIP stream and one set of scratch each

// What a frame of the brute-force / binned ray-trace paths writes besides the caller's planes: one set per stream, so
// that two frames in flight never share any of it.
// Small parameter blocks for the device (frame descriptors, ray origins): the words travel in the KERNEL ARGUMENTS of a one-
// workgroup kernel, which the launch copies before it returns -- ordered on the stream like any kernel and independent of when
// the runtime reads a pageable or stack source (hipMemcpyAsync from such memory leaves that to its staging policy).
struct UploadChunk { uint32_t w[768]; };
__global__ __launch_bounds__(256) void k_upload_words(const UploadChunk c, uint32_t *__restrict__ dst, int nwords)
{
    for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = c.w[i];
}
// ... and, in the same launch, up to two regions to zero (a pass's counters, the frame's hit counters): a fill of its own is a launch
// of its own, ~3 us of a single frame's latency each.
struct ZeroJob { uint32_t *a; int na; uint32_t *b; int nb; };
__global__ __launch_bounds__(256) void k_upload_words_zero(const UploadChunk c, uint32_t *__restrict__ dst, int nwords, const ZeroJob z)
{
    for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = c.w[i];
    for (int i = threadIdx.x; i < z.na; i += 256) z.a[i] = 0u;
    for (int i = threadIdx.x; i < z.nb; i += 256) z.b[i] = 0u;
}
hipError_t upload_small(void *dst, const void *src, size_t bytes, hipStream_t stream, const ZeroJob *zero = nullptr);
hipError_t upload_small(void *dst, const void *src, size_t bytes, hipStream_t stream, const ZeroJob *zero)
{
    if (zero) {
        // (the zero job rides on the first chunk)
        const uint32_t *w0 = static_cast<const uint32_t *>(src);
        UploadChunk c0;
        const int n0 = (int)std::min<size_t>(768, bytes / 4);
        memcpy(c0.w, w0, (size_t)n0 * 4);
        hipLaunchKernelGGL(k_upload_words_zero, dim3(1), dim3(256), 0, stream, c0, static_cast<uint32_t *>(dst), n0, *zero);
        if (bytes / 4 <= 768) return hipGetLastError();
        return upload_small(static_cast<uint32_t *>(dst) + 768, w0 + 768, bytes - 768 * 4, stream, nullptr);
    }
    const uint32_t *w = static_cast<const uint32_t *>(src);
    uint32_t *d = static_cast<uint32_t *>(dst);
    for (size_t off = 0, nw = bytes / 4; off < nw; off += 768) {
        UploadChunk c;
        const int n = (int)std::min<size_t>(768, nw - off);
        memcpy(c.w, w + off, (size_t)n * 4);
        hipLaunchKernelGGL(k_upload_words, dim3(1), dim3(256), 0, stream, c, d + off, n);
    }
    return hipGetLastError();
}

// The cost histogram of a pass leaves the device behind k_prep_select: its words go to a pinned copy the host reads later
// (weighted partition), and the device words are zero again for the next pass.
__global__ __launch_bounds__(SEL_HIST_MAX) void k_hist_out(uint32_t *__restrict__ hist, uint32_t *__restrict__ host_copy)
{
    host_copy[threadIdx.x] = hist[threadIdx.x];
    hist[threadIdx.x] = 0u;
}

struct RtScratch {
    OriginRow *d_cam_tab = nullptr;              // n rows (cam_tab_n)
    OriginRow *d_light_tab = nullptr;            // light_tab_lights x n rows
    int cam_tab_n = 0, light_tab_n = 0, light_tab_lights = 0;
    float *d_origins = nullptr;                  // (1 + MIRT_MAX_LIGHTS) x 3
    uint32_t *d_flags = nullptr;                 // [0] = unsafe flag
    // binned ray tracing: frame descriptors, per-bin offsets, the (bin, triangle) pair list and its sorted copy
    BinFrameDesc *d_frames = nullptr;
    uint32_t *d_bin_off = nullptr, *d_bin_counters = nullptr;
    uint32_t *d_entries = nullptr;               // triangle ids ordered by bin (the sorted pair values)
    uint32_t *d_pair_keys = nullptr, *d_pair_vals = nullptr, *d_sorted_keys = nullptr;   // unsorted pairs, sorted bin ids
    uint32_t *d_tmp_vals = nullptr;              // bucket sort: the pairs partitioned by bucket (keys go to d_sorted_keys)
    uint32_t *d_bucket = nullptr;                // bucket sort: counts | bases (+1) | cursors, cap_buckets each
    uint32_t cap_buckets = 0;
    bool bucket_dirty = false;                   // d_bucket may hold counts of a pass whose sort never ran
    // sizing the pair list without a host sync: the count of a frame is copied to pinned memory behind it and looked at by a
    // LATER frame of this stream; meanwhile the list is sized from the last count seen, with a device-side fallback if that
    // was too small (k_rt_trace2 then takes every triangle for every tile)
    uint32_t *h_count = nullptr;                 // pinned
    hipEvent_t ev_count = nullptr;
    bool count_pending = false;
    bool count_event_due = false;                // bin_pass published a count: the caller records ev_count behind the frame's last kernel
    bool have_known = false;
    uint32_t known_pairs = 0;
    uint32_t cap_bins = 0, cap_entries = 0;
    uint32_t cap_used = 0;                       // capacity the last binning pass told its kernels (== cap_entries outside tests)
    uint64_t bin_key = 0;
    uint32_t bin_entries = 0;                    // pairs of the current binning
    bool bin_key_valid = false;
    // lights that moved: this stream's own light-cube pass (rt_enqueue_binned), rows in the order of the pair list
    LightRow *d_light_rows = nullptr;
    uint32_t cap_light_rows = 0;
    // the frame's tile pairs ordered longest lists first (k_tile_order): ORDER_CLASSES segments of cap_order records
    TilePairRec *d_order = nullptr;
    uint32_t cap_order = 0;
    int last_bin_mode = -1;                      // what the last pass binned (camera alone / camera + n light cubes): a guessed
                                                 // list size only carries over between passes of the same kind
    // k_prep_select: the triangles the frame may see (indices, sel_n slots) and the two counters its passes use in turn (the pass
    // that counts into one zeroes the other: d_bin_counters[SEL_COUNT0 + parity])
    uint32_t *d_sel = nullptr;
    int sel_n = 0;
    int sel_parity = 0;
    // k_select_faces: per face of the light cubes this stream bins (its own frames' moving lights, or the shared cube's build) the
    // triangles the face can see -- list i at d_face_sel + i * n -- and the lists' lengths
    uint32_t *d_face_sel = nullptr;
    size_t cap_face_sel = 0;                     // slots
    uint32_t *d_face_counts = nullptr;           // 6 * MIRT_MAX_LIGHTS words
    // the cost histogram of the whole frame (weighted partition): device words, and where they travel for the host to read --
    // HIST_RING pinned copies taken in turn, an event behind each
    uint32_t *d_hist = nullptr;
    uint32_t *h_hist = nullptr;                  // pinned: HIST_RING x SEL_HIST_MAX words
    hipEvent_t ev_hist[4] = {};
    uint64_t hist_key[4] = {};                   // what frame (view, scene) each copy belongs to; 0 = none
    int hist_rows[4] = {}, hist_shift[4] = {};
    int hist_next = 0;
};
constexpr int SEL_COUNT0 = 80;                   // word of d_bin_counters where the two selection counters start
constexpr int HIST_RING = 4;

// The light-cube bins of the binned ray tracer: they depend on the scene and the light positions only, not on the camera,
// so they are built once per (scene version, light positions, grid) and shared by the frames of both streams.
struct LightCache {
    bool valid = false;
    uint64_t key = 0;                            // scene version + light positions (not the grid)
    int cube_bins = 0;                           // bins per face side of the tables held
    uint64_t track_key = 0;                      // the lights of the most recent binned frame ...
    int stable = 0;                              // ... and for how many frames in a row they have been the same
    OriginRow *d_light_tab = nullptr;            // nl x n origin rows
    size_t cap_tab = 0;
    BinFrameDesc *d_frames = nullptr;            // 6 x nl frame descriptors
    uint32_t *d_off = nullptr;                   // nbins + 1
    uint32_t cap_bins = 0, nbins = 0;
    LightRow *d_rows = nullptr;                  // expanded candidates in key order
    uint32_t *d_row_tri = nullptr;               // the triangle of each row
    uint32_t cap_rows = 0, nrows = 0;
    int shells = 1;                              // depth shells per bin of the tables held
    float *d_origins = nullptr;                  // (1 + MIRT_MAX_LIGHTS) x 3
    uint32_t *d_counter = nullptr;               // pair counter of the build
};

struct Ctx {
    bool init = false;
    bool profiling = false;
    int device = -1;
    int cu_count = 256;                          // multiProcessorCount of the device
    hipStream_t stream = nullptr;                // the stream of the current call (one of streams[])
    hipStream_t streams[MAX_FLIGHT] = {};
    int in_flight = 1;                           // frames that may be in flight at once (mirt_set_frames_in_flight)
    uint64_t frame_no = 0;                       // device calls so far
    int si = 0;                                  // index of the stream of the current / most recent call: calls take the streams in turn
    hipEvent_t ev_order[MAX_FLIGHT] = {};        // one per stream: orders work of one stream after what another has queued so far
    // a side stream per stream: the light-cube pass of a frame whose camera AND lights moved runs there, beside the camera's pass
    // (two latency-bound chains that share nothing until the trace kernel)
    hipStream_t aux[MAX_FLIGHT] = {};
    hipEvent_t ev_fork[MAX_FLIGHT] = {}, ev_join[MAX_FLIGHT] = {};
    // profiling events: one set per stream, so that the times of a frame survive the frames that follow it on the other streams
    // (mirt_get_previous_kernel_ms: the frame before the last one overlapped its neighbours on both sides)
    hipEvent_t ev_sets[MAX_FLIGHT][EV_COUNT] = {};
    bool ev_used_sets[MAX_FLIGHT][8] = {};
    bool call_timed_sets[MAX_FLIGHT] = {};
    int ev_cur = 0;                              // the set of the current / most recent call
    hipEvent_t *ev = ev_sets[0];
    bool *ev_used = ev_used_sets[0];

    // scene
    int n = 0;
    float *d_tris = nullptr;
    uint8_t *d_culled = nullptr;                 // isCulled flags: one copy of n per stream, [i * n, (i + 1) * n) for frames on streams[i]
    int culled_latest = 0;                       // which copy the most recent cull call wrote (mirt_scene_get_culled reads it)
    uint64_t culled_ver[MAX_FLIGHT] = {};        // what each copy holds: the number of the cull call (or upload) it comes from
    uint64_t cull_calls = 0;
    hipEvent_t ev_cull_read[MAX_FLIGHT] = {};    // per stream: its last copy OUT of another stream's flags has been made ...
    uint32_t cull_read_src[MAX_FLIGHT] = {};     // ... bit c: the stream has copied out of copy c since a cull step into c last waited for it
                                                 // (the event is re-recorded behind every such copy, and a stream runs in order: waiting
                                                 // for the latest record covers every earlier read of that stream)
    RtScratch rt[MAX_FLIGHT];                    // per-stream tables of the non-tile ray-trace paths (frames in flight)
    RtScratch rt_lt[MAX_FLIGHT];                 // per stream: the pair lists, offsets and rows of a LIGHT-cube pass -- the cubes of lights
                                                 // that move (binned by the frame) and the scratch of the shared cube's build -- apart from the
                                                 // camera's, so that either pass is kept while only the other one's inputs change
    GeoRow *d_geo = nullptr;                     // n geometry rows (built by mirt_scene_upload)
    ShadeRow *d_shade = nullptr;                 // n shading rows (likewise)
    float bbox_lo[3] = { 0, 0, 0 }, bbox_hi[3] = { 0, 0, 0 };   // the scene's bounding box (host side, mirt_scene_upload)
    LightCache lc;
    unsigned long long *d_hits = nullptr;        // the hit-counter buffer of the current frame (one of d_hits2)
    unsigned long long *d_hits2[2 * MAX_FLIGHT] = {};   // HIT_SHARDS sharded counters each (rt_common.hpp: count_hits): two per stream, [si + MAX_FLIGHT * toggle]
    bool hits_clean[2 * MAX_FLIGHT] = {};        // buffer is all zero (the tile kernel clears its stream's other one itself)
    int hits_cur = 0;
    float4 *d_tile_tab[MAX_FLIGHT] = {};         // per-stream tables of the tile ray tracer (k_tile_tables)
    int hits_tog[MAX_FLIGHT] = {};
    bool scene_finite = true;                    // all vertex coordinates below MIRT_SAFE_MAG
    uint64_t scene_version = 0;                  // bumped whenever the triangles change
    uint64_t cull_version = 0;                   // bumped whenever the cull flags change (rasteriser sizing only)
    int soft_samples = 1;                        // soft-shadow samples per light (1 = hard shadows)
    int aa = 1;                                  // realSamples of Draw(): AA_SAMPLES when AA_ENABLED, else 1
    int dof_k = 0;                               // DOF_KERNEL_SIZE when DOF_ENABLED, else 0
    float dof_focal = 0.0f;                      // FOCAL_LENGTH
    struct DofPlanes {                           // one set per stream (frames in flight)
        float *rgb = nullptr, *fd = nullptr;     // pixelColours / focalDistances of the band + halo
        uint32_t *xrgb = nullptr;                // unblurred words the render kernels emit (discarded)
        int32_t *index = nullptr;
        float *zinv = nullptr;
        size_t cap_px = 0;
    } dof[MAX_FLIGHT];
    int soft_npos = 0;
    float soft_pos[MIRT_MAX_LIGHTS * 3] = {};    // jittered light positions, [light*samples + i]

    // host surfaces the caller registered (mirt_surface_register): pinned + mapped, so the frame reaches them at link speed
    struct HostSurface { char *host = nullptr; char *dev = nullptr; size_t bytes = 0; } surf[4];

    // staging for the host-buffer entry points
    void *d_xrgb = nullptr, *d_rgb = nullptr, *d_index = nullptr, *d_zinv = nullptr, *d_pos = nullptr;
    size_t cap_px = 0;
    // ... and for the asynchronous ones: one XRGB plane per stream -- the frame that reuses a plane is queued on the stream
    // whose copy engine read it last, so the render is ordered after that copy whatever other calls came in between
    void *d_async[MAX_FLIGHT] = {};
    size_t async_cap_px = 0;
    RasterScratch raster[MAX_FLIGHT];            // one set of rasteriser scratch per stream (frames in flight)

    // several GPUs: this process's place among the ranks that shard a frame, and its band buffers (two: the gather of one
    // batch overlaps the render of the next)
    Comm *comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_rendered = nullptr, ev_sent[2] = { nullptr, nullptr };
    char *d_band[2] = { nullptr, nullptr };
    size_t band_bytes[2] = { 0, 0 };
    int band_slot = 0;
    int strip_rows = 0;                          // partition of a sharded frame: 0 = contiguous bands, > 0 = interleaved strips of that many rows,
                                                 // MIRT_PARTITION_WEIGHTED = bands of equal estimated cost (mirt_set_partition)
    bool want_hist = false;                      // binned ray-traced frames leave their cost histogram (mirt_set_cost_histogram, or the weighted partition)
    uint64_t shard_calls = 0;                    // sharded calls so far: what a cost histogram is filed under
    bool hist_taken = false;                     // the current sharded call has filed its histogram (the first binned pass of a call does)
    bool in_sharded = false;

    // statistics of the last call
    mirt_stats stats = {};
    bool stats_pending = false;
    bool raster_since_sync = false;              // rasteriser frames were queued since the last mirt_sync (overflow check there)
    bool call_timed = false;                     // the last call recorded its start / end events (profiling was on)
    hipStream_t stats_stream = nullptr;          // the stream the last call ran on
    uint64_t pending_primary = 0;
    int pending_nlights = 0;
    bool pending_is_rt = false;
    bool pending_counted = false;                // the kernel counted its executed tests itself (tile / binned)
    bool pending_empty = false;                  // the last ray-trace call rendered no rows (no counters to read)
    const uint32_t *stats_sel_count = nullptr;   // binned frame that ran a pass: where its selection count is (device)
};

Ctx g;

int need_init()
{
    if (!g.init) return fail(MIRT_ERR_NOT_INITIALISED, "mirt_init has not been called (or failed)");
    return MIRT_OK;
}

template <typename T>
int dev_realloc(T **p, size_t count)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) return MIRT_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
    if (e != hipSuccess) { *p = nullptr; return fail(MIRT_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e)); }
    return MIRT_OK;
}

int ensure_staging(size_t px, bool rgb, bool index, bool zinv, bool pos = false)
{
    // All staging planes share ONE capacity (g.cap_px pixels): a plane that is first needed by a small frame must
    // still be big enough for every frame size the other planes were already grown to.
    if (px > g.cap_px) {
        for (void **p : { &g.d_xrgb, &g.d_rgb, &g.d_index, &g.d_zinv, &g.d_pos }) { if (*p) (void)hipFree(*p); *p = nullptr; }
        g.cap_px = px;
    }
    auto grow = [&](void **p, size_t bytes_per_px) -> int {
        if (*p) return MIRT_OK;
        const size_t bytes = g.cap_px * bytes_per_px;
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) { *p = nullptr; return fail(MIRT_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
        return MIRT_OK;
    };
    int rc;
    if ((rc = grow(&g.d_xrgb, 4))) return rc;
    if (rgb && (rc = grow(&g.d_rgb, 12))) return rc;
    if (index && (rc = grow(&g.d_index, 4))) return rc;
    if (zinv && (rc = grow(&g.d_zinv, 4))) return rc;
    if (pos && (rc = grow(&g.d_pos, 12))) return rc;
    return MIRT_OK;
}

bool finite_below(const float *p, int n, float lim)
{
    for (int i = 0; i < n; i++) if (!(fabsf(p[i]) < lim)) return false;
    return true;
}

int check_view(const mirt_view *v, const mirt_light *lights, int nlights, const float *indirect)
{
    if (!v || !indirect) return fail(MIRT_ERR_INVALID_ARGUMENT, "view / indirect must not be NULL");
    if (v->width < 1 || v->height < 1 || v->width > 32768 || v->height > 32768)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "frame size %dx%d out of range [1,32768]", v->width, v->height);
    if (nlights < 0 || nlights > MIRT_MAX_LIGHTS) return fail(MIRT_ERR_INVALID_ARGUMENT, "nlights %d out of range [0,%d]", nlights, MIRT_MAX_LIGHTS);
    if (nlights > 0 && !lights) return fail(MIRT_ERR_INVALID_ARGUMENT, "lights must not be NULL when nlights > 0");
    return MIRT_OK;
}

void k_begin(int k) { if (g.profiling) { (void)hipEventRecord(g.ev[EV_K0 + 2 * k], g.stream); g.ev_used[k] = true; } }
// Waits for every call enqueued so far (all streams).
hipError_t sync_all()
{
    hipError_t e = hipSuccess;
    for (int i = 0; i < MAX_FLIGHT; i++)
        if (g.streams[i]) { const hipError_t r = hipStreamSynchronize(g.streams[i]); if (r != hipSuccess) e = r; }
    for (int i = 0; i < MAX_FLIGHT; i++)
        if (g.aux[i]) { const hipError_t r = hipStreamSynchronize(g.aux[i]); if (r != hipSuccess) e = r; }
    if (g.comm_stream) { const hipError_t r = hipStreamSynchronize(g.comm_stream); if (r != hipSuccess) e = r; }
    return e;
}

void k_end(int k) { if (g.profiling) (void)hipEventRecord(g.ev[EV_K0 + 2 * k + 1], g.stream); }

// The stream the NEXT device call will take: calls take the in_flight streams in turn.
int next_si() { return g.in_flight > 1 ? (g.si + 1) % g.in_flight : 0; }

// Every device call starts here.  With several frames in flight consecutive calls take the streams in turn, so frame i+1 is
// dispatched -- and its kernels run, where the device has room -- while frame i still drains: no dispatch gap, no idle tail,
// and the latency-bound chains of consecutive frames (binning, sort, trace; vertex, edges, fragments, resolve) fill each
// other's gaps.  A frame reads the scene and writes the caller's planes plus its OWN stream's scratch (origin tables, bins,
// raster keys, depth-of-field planes, counters), so frames need no ordering among themselves; frames i and i + in_flight,
// which a caller cycling through in_flight sets of planes gives the same planes, share a stream.
void call_begin()
{
    g.si = next_si();
    g.stream = g.streams[g.si];
    g.frame_no++;
    (void)hipGetLastError();                     // drop a stale error of another HIP user in this thread (torch polls events:
                                                 // hipErrorNotReady) so that the launch checks below report our own launches only
    memset(&g.stats, 0, sizeof g.stats);
    g.stats_sel_count = nullptr;
    g.ev_cur = g.si;
    g.ev = g.ev_sets[g.ev_cur];
    g.ev_used = g.ev_used_sets[g.ev_cur];
    memset(g.ev_used, 0, sizeof g.ev_used_sets[0]);
    // the call's own start / end events only when profiling is on: an event record costs ~2.7 us of host time, a quarter of
    // a 500 x 500 Cornell frame (12.9 -> 7.x us per frame without the two of them)
    g.call_timed = g.profiling;
    g.call_timed_sets[g.ev_cur] = g.call_timed;
    if (g.call_timed) (void)hipEventRecord(g.ev[EV_CALL0], g.stream);
}
void call_end() { if (g.call_timed) (void)hipEventRecord(g.ev[EV_CALL1], g.stream); g.stats_stream = g.stream; g.stats_pending = true; }

// ---- ray tracer --------------------------------------------------------------------------------------

// The camera ray family negD = -(R0*(x - W/2) + R1*(y - H/2) + R2*f) as a bin frame: (u, v) = pixel (x, y), bins =
// 8x8-pixel tiles; also carries the inverse map for the bounding boxes (rt_binned.hpp).
BinFrameDesc make_camera_frame(const mirt_view *view, int y0, int y1, int aa)
{
    const int W = view->width, H = view->height;
    BinFrameDesc c;
    memset(&c, 0, sizeof c);
    {
        const float *R = view->rot;                       // column-major: column j = R[3j..3j+2]
        const float hw = (float)W / 2.0f, hh = (float)H / 2.0f;
        for (int i = 0; i < 3; i++) {
            c.Pu[i] = -R[0 + i];
            c.Pv[i] = -R[3 + i];
            c.P0[i] = -(R[6 + i] * view->focal - R[0 + i] * hw - R[3 + i] * hh);
        }
        float dm = 0.0f;
        for (int i = 0; i < 3; i++)
            dm = fmaxf(dm, fabsf(R[0 + i]) * (hw + 1.0f) + fabsf(R[3 + i]) * (hh + 1.0f) + fabsf(R[6 + i]) * fabsf(view->focal));
        c.dmax = dm;
        // inverse map for the bounding boxes: h = R^-1 (P - S) = lambda * (x - W/2, y - H/2, f), so with g = S - P
        //   w = -(R^-1 row 2 . g) / f,  u = (-(R^-1 row 0 . g) + (W/2) f w / f ... ) -> rows below; computed in double
        {
            double M[9], inv[9];
            for (int i = 0; i < 9; i++) M[i] = R[i];
#define MM(cc, rr) M[(cc) * 3 + (rr)]
            const double det = MM(0, 0) * (MM(1, 1) * MM(2, 2) - MM(2, 1) * MM(1, 2)) - MM(1, 0) * (MM(0, 1) * MM(2, 2) - MM(2, 1) * MM(0, 2)) +
                               MM(2, 0) * (MM(0, 1) * MM(1, 2) - MM(1, 1) * MM(0, 2));
            // inv is row-major here: inv[r*3+c] = (R^-1)(r, c)
            inv[0] = (MM(1, 1) * MM(2, 2) - MM(2, 1) * MM(1, 2)) / det; inv[1] = -(MM(1, 0) * MM(2, 2) - MM(2, 0) * MM(1, 2)) / det; inv[2] = (MM(1, 0) * MM(2, 1) - MM(2, 0) * MM(1, 1)) / det;
            inv[3] = -(MM(0, 1) * MM(2, 2) - MM(2, 1) * MM(0, 2)) / det; inv[4] = (MM(0, 0) * MM(2, 2) - MM(2, 0) * MM(0, 2)) / det; inv[5] = -(MM(0, 0) * MM(2, 1) - MM(2, 0) * MM(0, 1)) / det;
            inv[6] = (MM(0, 1) * MM(1, 2) - MM(1, 1) * MM(0, 2)) / det; inv[7] = -(MM(0, 0) * MM(1, 2) - MM(1, 0) * MM(0, 2)) / det; inv[8] = (MM(0, 0) * MM(1, 1) - MM(1, 0) * MM(0, 1)) / det;
#undef MM
            const bool ok = std::isfinite(det) && det != 0.0 && view->focal != 0.0f;
            for (int i = 0; i < 3; i++) {
                const double rwd = ok ? -inv[6 + i] / (double)view->focal : 0.0;       // w = h.z / f, h = -R^-1 g
                c.rw[i] = (float)rwd;
                c.ru[i] = (float)(ok ? -inv[0 + i] + (double)hw * rwd : 0.0);          // u*w = h.x + (W/2) w
                c.rv[i] = (float)(ok ? -inv[3 + i] + (double)hh * rwd : 0.0);
            }
        }
        memcpy(c.S, view->pos, 12);
        c.ulo = 0.0f; c.vlo = 0.0f; c.du = (float)BIN_TILE; c.dv = (float)BIN_TILE;
        // bin i covers the rays of pixels 8i .. 8i+7: exactly their centres, or with supersampling half a pixel around them
        c.pad_lo = aa > 1 ? -0.5f : 0.0f; c.pad_hi = aa > 1 ? -0.5f : -1.0f;
        c.nbu = (W + BIN_TILE - 1) / BIN_TILE; c.nbv = (H + BIN_TILE - 1) / BIN_TILE;
        c.j0 = y0 / BIN_TILE; c.j1 = (y1 + BIN_TILE - 1) / BIN_TILE;
        c.base = 0; c.tab = 0;
    }
    return c;
}

// ---- binned ray tracing ---------------------------------------------------------------------------------------------

// Makes room for `cap` (bin, triangle) pairs in a stream's pair list, its sorted copy and the sort's scratch.
int ensure_pairs(RtScratch &S, size_t cap)
{
    int r;
    if ((r = dev_realloc(&S.d_entries, cap)) || (r = dev_realloc(&S.d_pair_keys, cap)) || (r = dev_realloc(&S.d_pair_vals, cap)) ||
        (r = dev_realloc(&S.d_sorted_keys, cap)) || (r = dev_realloc(&S.d_tmp_vals, cap))) { S.cap_entries = 0; return r; }
    S.cap_entries = (uint32_t)cap;
    return MIRT_OK;
}

// Most sort keys (bin * depth shells + shell) one binning pass may use: the two-level counting sort keeps one LDS counter per
// bucket of at most 1024 keys (bin_bucket_sort.hip).  The callers choose their grids and shell counts to stay below it.
constexpr uint32_t BIN_MAX_KEYS = BUCKET_SORT_MAX_BUCKETS * 1024u - 1u;

// Picks up the pair count an earlier frame of the stream has published (pinned word + event), if it has landed.
void poll_pair_count(RtScratch &S)
{
    if (S.count_pending && hipEventQuery(S.ev_count) == hipSuccess) {
        S.known_pairs = *S.h_count; S.have_known = true; S.count_pending = false;
    }
    (void)hipGetLastError();                                 // (hipErrorNotReady of the query is not an error)
}

// One binning pass on g.stream: (key, triangle) pairs of `bs`' frames into S' pair list, ordered by key into S.d_entries /
// S.d_sorted_keys, offsets into bin_off.  `counter` (device, zeroed by the caller's previous kernel) receives the pair
// count.  The list is sized from a count only the device knows: when `fresh` it is read back (4 bytes + one sync of this
// stream) and the pass repeated if the list was too small; otherwise *npairs, the count of the identical pass before, holds.
// A pass that may not read back (`may_guess`) sizes the list from the count an earlier pass published and publishes its own;
// a list that turns out too small makes the frame's kernels take the brute-force path (k_rt_trace2) and the NEXT pass grow it.
int bin_pass(RtScratch &S, BinSet bs, const OriginRow *cam_tab, const OriginRow *light_tab, uint32_t *counter, uint32_t *bin_off,
             bool fresh, uint32_t *npairs, bool may_guess = false)
{
    int rc;
    poll_pair_count(S);                                      // a count an earlier frame left behind?
    // A pass identical to the one before it (same view, same scene) normally reuses that pass's count without looking; but if
    // that pass was itself a guess and its published count shows the list was too small, the frame fell back to brute force
    // and so would every later frame of this view: treat it as fresh again so that the list grows.
    if (!fresh && may_guess && S.have_known && S.known_pairs > S.cap_used) fresh = true;
    const bool guess = fresh && may_guess && S.have_known;
    if (!S.d_entries || !S.cap_entries) {
        // first capacity of the pair list (grown on demand below); MIRT_BIN_INITIAL_PAIRS lets a test start small
        static const size_t initial = [] { const char *e = getenv("MIRT_BIN_INITIAL_PAIRS"); long v = e ? atol(e) : 0; return v > 0 ? (size_t)v : (size_t)1 << 20; }();
        if ((rc = ensure_pairs(S, initial))) return rc;
    }
    if (bs.nbins > BIN_MAX_KEYS) return fail(MIRT_ERR_INVALID_ARGUMENT, "binning: %u sort keys exceed the %u the bucket sort holds", bs.nbins, BIN_MAX_KEYS);
    // workgroups striding over the (256-triangle chunk, frame) work items: 8 per CU (52 KiB of LDS and 512 threads each, 3 resident; 1 M
    // triangles at 8K: 4.06 -> 3.53 ms per frame against 3 per CU)
    bs.chunk_tris = 256;                                // (64 measured slower on the 100 k soup: 86 vs 74 us for the whole binning, more flushes)
    const dim3 bin_grid((unsigned)std::min<long long>((long long)((g.n + bs.chunk_tris - 1) / bs.chunk_tris) * bs.nframes, (long long)g.cu_count * 8));
    bs.counters = counter;
    // order the pairs by key with the two-level counting sort (bin_bucket_sort.hip: k_bin_pairs counts the pairs per bucket,
    // two more launches sort)
    const uint32_t nbuckets = bucket_sort_buckets(bs.nbins);
    if (nbuckets + 1 > S.cap_buckets) {
        S.cap_buckets = 0;
        if ((rc = dev_realloc(&S.d_bucket, (size_t)3 * (nbuckets + 1)))) return rc;
        HIP_TRY(hipMemsetAsync(S.d_bucket, 0, sizeof(uint32_t) * 3 * (nbuckets + 1), g.stream));
        S.cap_buckets = nbuckets + 1;
        S.bucket_dirty = false;
    }
    uint32_t *bcnt = S.d_bucket, *bbase = S.d_bucket + S.cap_buckets, *bcur = S.d_bucket + 2 * (size_t)S.cap_buckets;
    bs.bucket_cnt = bcnt; bs.nbuckets = nbuckets; bs.bucket_shift = bucket_sort_shift(bs.nbins);
    const size_t bin_lds = (size_t)nbuckets * sizeof(uint32_t);
    {   // k_bin_pairs: ~52 KB of static LDS + up to 32 KB of bucket counters: past the 64 KB a launch may use by default
        static const bool once = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_pairs<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bin_pairs<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
            return true; }();
        (void)once;
    }
    // workgroups of 256 threads where the frame's kernels overlap with its neighbours' -- three or four frames in flight, a scene small
    // enough for that to matter --, of 512 for the large scenes and for the frame that runs alone, whose latency they serve (rt_binned.hip)
    static const int bin_wg_env = [] { const char *e = getenv("MIRT_BIN_WG"); return e ? atoi(e) : 0; }();
    const int bin_wg = (bin_wg_env == 256 || bin_wg_env == 512) ? bin_wg_env : ((g.n < 400000 && g.in_flight >= 3) ? 256 : 512);
    if (guess) {
        // room for half as many pairs again as the last frame seen produced; growing needs this stream idle (rare)
        const size_t want = (size_t)S.known_pairs + S.known_pairs / 2 + 4096;
        if (want > S.cap_entries) {
            HIP_TRY(hipStreamSynchronize(g.stream));
            if ((rc = ensure_pairs(S, want + want / 4))) return rc;
        }
    }
    bool publish_count = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        // MIRT_TEST_PAIR_CAP (tests only): a guessed list pretends to be this small, so that the overflow path runs
        static const uint32_t test_cap = [] { const char *e = getenv("MIRT_TEST_PAIR_CAP"); long v = e ? atol(e) : 0; return v > 0 ? (uint32_t)v : 0u; }();
        S.cap_used = (guess && test_cap && test_cap < S.cap_entries) ? test_cap : S.cap_entries;
        BinPairs pairs = { S.d_pair_keys, S.d_pair_vals, S.cap_used };
        bs.entries = S.d_entries; bs.cap_entries = S.cap_used;
        if (attempt) HIP_TRY(hipMemsetAsync(counter, 0, 4, g.stream));
        if (attempt || S.bucket_dirty) HIP_TRY(hipMemsetAsync(S.d_bucket, 0, sizeof(uint32_t) * 3 * (size_t)S.cap_buckets, g.stream));
        S.bucket_dirty = true;                               // bucket counts pending until k_bs_local has consumed them
        if (bin_wg == 256) hipLaunchKernelGGL(k_bin_pairs<256>, bin_grid, dim3(256), bin_lds, g.stream, g.d_tris, cam_tab, light_tab, g.n, bs, pairs);
        else hipLaunchKernelGGL(k_bin_pairs<512>, bin_grid, dim3(512), bin_lds, g.stream, g.d_tris, cam_tab, light_tab, g.n, bs, pairs);
        if (!fresh) break;
        if (guess) {
            // no sync: k_bs_scatter stores the count into a pinned word behind the kernel and a later frame picks it up
            if (!S.h_count) {
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.h_count), 64, hipHostMallocDefault));
                HIP_TRY(hipEventCreateWithFlags(&S.ev_count, hipEventDisableTiming));
            }
            if (!S.count_pending) publish_count = true;
            *npairs = S.known_pairs;
            break;
        }
        uint32_t total = 0;
        HIP_TRY(hipStreamSynchronize(g.stream));
        HIP_TRY(hipMemcpy(&total, counter, 4, hipMemcpyDeviceToHost));
        *npairs = total;
        S.known_pairs = total; S.have_known = true;
        S.count_pending = false;                             // (a count still on its way belongs to an earlier pass, maybe of another kind)
        if (total <= S.cap_entries) break;
        if (attempt == 1) return fail(MIRT_ERR_HIP, "binning produced %u pairs twice with room for %u", total, S.cap_entries);
        if ((rc = ensure_pairs(S, (size_t)total + total / 8 + 4096))) return rc;
    }
#ifdef MIRT_BIN_STATS
    if (fresh) {
        uint32_t c[16];
        (void)hipMemcpy(c, counter, 64, hipMemcpyDeviceToHost);
        fprintf(stderr, "[mirt bin stats] flattened tests=%u max per work item=%u direct items=%u | huge: box valid=%u no box=%u (camera frame %u)\n", c[8], c[9], c[10], c[11], c[12], c[14]);
        fprintf(stderr, "[mirt bin stats] tris=%d frames=%d  pairs=%u  bins=%u | large items walked=%u level-1 rounds=%u level-2 steps=%u pairs=%u max steps/item=%u items>100 steps=%u\n",
                g.n, bs.nframes, *npairs, bs.nbins, c[2], c[3], c[4], c[5], c[6], c[7]);
        (void)hipMemset(counter + 2, 0, 56);
    }
#endif
    uint32_t *count_out = nullptr;
    if (publish_count) HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&count_out), S.h_count, 0));
    HIP_TRY(bucket_sort_pairs(S.d_pair_keys, S.d_pair_vals, counter, S.cap_used, *npairs, bs.nbins, S.d_sorted_keys, S.d_tmp_vals,
                              bcnt, bbase, bcur, bin_off, S.d_entries, g.cu_count, g.stream, count_out));
    S.bucket_dirty = false;                                  // k_bs_local leaves the counts and cursors zero
    // (the event that tells a later frame the count has landed is recorded by the caller BEHIND the frame's trace kernel: an event
    // record between two kernels of the chain is a barrier packet of its own, ~5 us of the single frame's latency)
    if (publish_count) S.count_event_due = true;
    return MIRT_OK;
}

// Key of what the light-cube bins depend on: the scene and the light positions.
uint64_t light_key_of(const float *origins, int nlights)
{
    uint64_t key = 0xcbf29ce484222325ull ^ g.scene_version;
    auto mix = [&](const void *p, size_t nb) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < nb; i++) { key ^= b[i]; key *= 0x100000001b3ull; } };
    mix(origins + 3, sizeof(float) * 3 * nlights); mix(&nlights, 4); mix(&g.n, 4);
    return key;
}

// Nearest and farthest distance from `pos` to the scene's bounding box: the range the depth shells of a ray family divide.
bool shell_range(const float *pos, double *dn, double *df)
{
    double n2 = 0.0, f2 = 0.0;
    for (int c = 0; c < 3; c++) {
        const double p = pos[c], lo = g.bbox_lo[c], hi = g.bbox_hi[c];
        const double near = p < lo ? lo - p : (p > hi ? p - hi : 0.0), far = std::max(std::fabs(p - lo), std::fabs(p - hi));
        n2 += near * near; f2 += far * far;
    }
    *dn = std::sqrt(n2); *df = std::sqrt(f2);
    return std::isfinite(*dn) && std::isfinite(*df) && *df > *dn;
}

// Frame descriptors of the light cubes: six faces of B x B bins around every light position, every bin's list ordered in
// `shells` depth shells of the candidates' `near` bound (sort key = (base + bin) * shells + shell; `base_bins` = where light 0's
// face 0 starts, in bins of `shells` keys).  A shadow ray walks only the shells up to the one its 0.99 r falls into (k_rt_trace2).
void fill_light_frames(BinFrameDesc *frames, const RtFrame &f, int nlights, int cube_bins, int shells, uint32_t base_bins)
{
    memset(frames, 0, sizeof(BinFrameDesc) * 6 * nlights);
    for (int k = 0; k < nlights; k++) {
        double dn = 0.0, df = 0.0;
        const bool okr = shell_range(f.lpos[k], &dn, &df);
        for (int face = 0; face < 6; face++) {
            BinFrameDesc &d = frames[k * 6 + face];
            const int ax = face >> 1;
            d.P0[ax] = (face & 1) ? -1.0f : 1.0f;         // negD ~ s*e_k + u*e_(k+1) + v*e_(k+2)
            d.Pu[(ax + 1) % 3] = 1.0f;
            d.Pv[(ax + 2) % 3] = 1.0f;
            d.rw[ax] = d.P0[ax]; d.ru[(ax + 1) % 3] = 1.0f; d.rv[(ax + 2) % 3] = 1.0f;   // g = m*(s e_k + u e_k1 + v e_k2)
            memcpy(d.S, f.lpos[k], 12);                       // light position k (jittered sample with soft shadows)
            d.dmax = 2.0f;
            d.ulo = -1.0f; d.vlo = -1.0f; d.du = 2.0f / (float)cube_bins; d.dv = 2.0f / (float)cube_bins;
            d.pad_lo = -3.814697265625e-06f; d.pad_hi = 3.814697265625e-06f;
            d.nbu = cube_bins; d.nbv = cube_bins; d.j0 = 0; d.j1 = cube_bins;
            d.base = base_bins; d.tab = 1 + k;
            // every face of every light carries `shells` keys per bin (the key layout needs one count for all); a light whose
            // range is degenerate puts everything into shell 0
            d.nshell = shells;
            d.shell_d0 = (float)dn;
            d.shell_iw = okr ? (float)(shells / (df - dn)) : 0.0f;
            base_bins += (uint32_t)(cube_bins * cube_bins);
        }
    }
}

// Depth shells per light-cube bin: as many as the sort's key space allows, at most 16 (a bin's list grows with the square of
// the distance from the light; 16 shells leave a ray at a quarter of the scene's depth ~2 % of it).
int light_shells_for(int nlights, int cube_bins, uint32_t keys_in_front)
{
    static const int env = [] { const char *e = getenv("MIRT_LIGHT_SHELLS"); return e ? atoi(e) : 0; }();
    const long long bins = 6ll * cube_bins * cube_bins * std::max(nlights, 1);
    int ns = (env >= 1 && env <= 64) ? env : 16;
    while (ns > 1 && bins * ns + keys_in_front + 64 > (long long)BIN_MAX_KEYS) ns >>= 1;
    return ns;
}

constexpr size_t LIGHT_COUNTER_BYTES = sizeof(uint32_t) * (128 + 6 * MIRT_MAX_LIGHTS);
// Room for the face lists of `nlights` light cubes (k_select_faces) in a stream's LIGHT scratch set; the stream must be idle when they grow.
int ensure_face_lists(RtScratch &S, int nlights)
{
    int rc;
    if (!S.d_bin_counters) {
        // the light pass's counters and the face lists' lengths in ONE block (a pass zeroes it with one fill): words 0..127 as in the
        // camera's block, 128.. the face counts
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.d_bin_counters), LIGHT_COUNTER_BYTES));
        HIP_TRY(hipMemsetAsync(S.d_bin_counters, 0, LIGHT_COUNTER_BYTES, g.stream));
        S.d_face_counts = S.d_bin_counters + 128;
    }
    const size_t want = (size_t)6 * (size_t)nlights * (size_t)g.n;
    if (want > S.cap_face_sel) {
        HIP_TRY(hipStreamSynchronize(g.stream));
        S.cap_face_sel = 0;
        if ((rc = dev_realloc(&S.d_face_sel, want))) return rc;
        S.cap_face_sel = want;
    }
    return MIRT_OK;
}

// The SHARED light-cube bins and their expanded rows, for lights that stand still: built on g.stream as a barrier call -- the
// frames of both streams read the tables -- whenever the scene, a light position or the grid differs from what is held.  (Lights
// that just moved do not come here: rt_enqueue_binned bins their cubes together with the camera frame, on the frame's own stream.)
int light_cache_ensure(RtScratch &S, const RtFrame &f, const float *origins, int nlights, int cube_bins)
{
    int rc;
    LightCache &C = g.lc;
    const uint64_t key = light_key_of(origins, nlights);
    if (C.valid && C.key == key && C.cube_bins == cube_bins) return MIRT_OK;
    C.valid = false;
    for (int o = 0; o < g.in_flight; o++)                  // frames of the other streams may still read the old tables
        if (o != g.si) {
            HIP_TRY(hipEventRecord(g.ev_order[o], g.streams[o]));
            HIP_TRY(hipStreamWaitEvent(g.stream, g.ev_order[o], 0));
        }
    const int shells = light_shells_for(nlights, cube_bins, 0u);
    const uint32_t per_light = 6u * (uint32_t)(cube_bins * cube_bins) * (uint32_t)shells, nkeys = per_light * (uint32_t)nlights;
    if ((size_t)nlights * g.n > C.cap_tab) {
        C.cap_tab = 0;
        if ((rc = dev_realloc(&C.d_light_tab, (size_t)nlights * g.n))) return rc;
        C.cap_tab = (size_t)nlights * g.n;
    }
    if (nkeys + 1 > C.cap_bins) {
        C.cap_bins = 0;
        if ((rc = dev_realloc(&C.d_off, (size_t)nkeys + 1))) return rc;
        C.cap_bins = nkeys + 1;
    }
    if (!C.d_frames) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&C.d_frames), sizeof(BinFrameDesc) * 6 * MIRT_MAX_LIGHTS));
    if (!C.d_origins) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&C.d_origins), sizeof(float) * 3 * (1 + MIRT_MAX_LIGHTS)));
    if (!C.d_counter) { HIP_TRY(hipMalloc(reinterpret_cast<void **>(&C.d_counter), 512)); HIP_TRY(hipMemsetAsync(C.d_counter, 0, 512, g.stream)); }   // (k_prep_origin zeroes words 0 and 16..79 of a pass's counter block)   // (ON the stream: see zero-fill note at S.d_bin_counters)
    C.nbins = nkeys;
    C.nrows = 0;
    C.shells = shells;
    if (nlights > 0) {
        BinFrameDesc frames[6 * MIRT_MAX_LIGHTS];
        fill_light_frames(frames, f, nlights, cube_bins, shells, 0u);
        HIP_TRY(upload_small(C.d_frames, frames, sizeof(BinFrameDesc) * 6 * nlights, g.stream));
        HIP_TRY(upload_small(C.d_origins, origins, sizeof(float) * 3 * (1 + nlights), g.stream));
        // the lights' origin rows, and per face the triangles it can see (k_select_faces); the build's pair counter is zeroed on the way
        if ((rc = ensure_face_lists(S, nlights))) return rc;
        HIP_TRY(hipMemsetAsync(S.d_face_counts, 0, sizeof(uint32_t) * 6 * nlights, g.stream));
        HIP_TRY(hipMemsetAsync(C.d_counter, 0, 512, g.stream));
        S.count_event_due = false;
        hipLaunchKernelGGL(k_select_faces, dim3((unsigned)std::min<long long>(((long long)g.n + 1023) / 1024, (long long)g.cu_count), nlights), dim3(1024), 0, g.stream,
                           g.d_tris, g.n, C.d_origins, C.d_frames, C.d_light_tab, S.d_face_sel, (uint32_t)g.n, S.d_face_counts);
        BinSet bs;
        memset(&bs, 0, sizeof bs);
        bs.frames = C.d_frames; bs.nframes = 6 * nlights; bs.nbins = nkeys; bs.bin_off = C.d_off;
        bs.face_lists = S.d_face_sel; bs.face_counts = S.d_face_counts; bs.face_stride = (uint32_t)g.n;
        uint32_t npairs = 0;
        if ((rc = bin_pass(S, bs, nullptr, C.d_light_tab, C.d_counter, C.d_off, true, &npairs))) return rc;
        S.count_event_due = false;                           // (a fresh pass without a guess reads its count back: nothing was published)
        if (npairs > C.cap_rows) {
            C.cap_rows = 0;
            if ((rc = dev_realloc(&C.d_rows, (size_t)npairs + npairs / 8 + 1024))) return rc;
            if ((rc = dev_realloc(&C.d_row_tri, (size_t)npairs + npairs / 8 + 1024))) return rc;
            C.cap_rows = npairs + npairs / 8 + 1024;
        }
        C.nrows = npairs;
        if (npairs)
            hipLaunchKernelGGL(k_expand_light_rows, dim3((unsigned)std::min<uint32_t>((npairs + 255) / 256, 4096u)), dim3(256), 0, g.stream,
                               C.d_off, S.d_entries, nlights, per_light, C.d_light_tab, g.n, C.d_rows, (const uint32_t *)nullptr, 0u, C.d_row_tri);
        HIP_TRY(hipGetLastError());
        S.bin_key_valid = false;                             // the stream's pair list now holds the light pass
        S.last_bin_mode = -1;
    } else {
        HIP_TRY(hipMemsetAsync(C.d_off, 0, 4, g.stream));
    }
    if (g.in_flight > 1) {                                   // later frames of the other streams wait for the build
        HIP_TRY(hipEventRecord(g.ev_order[g.si], g.stream));
        for (int o = 0; o < g.in_flight; o++)
            if (o != g.si) HIP_TRY(hipStreamWaitEvent(g.streams[o], g.ev_order[g.si], 0));
    }
    C.key = key;
    C.cube_bins = cube_bins;
    C.valid = true;
    return MIRT_OK;
}

// Depth shells of the camera bins for a frame of `tiles` bins (the tiles' lists come out of the sort roughly front to back).
int camera_shells_for(long long tiles)
{
    static const int shells_env = [] { const char *e = getenv("MIRT_CAM_SHELLS"); return e ? atoi(e) : 0; }();
    int ns = (int)std::min<long long>(8, std::max<long long>(1, (4ll << 20) / std::max<long long>(tiles, 1)));
    if (shells_env >= 1 && shells_env <= 64) ns = shells_env;
    while (ns > 1 && tiles * ns + 64 > (long long)BIN_MAX_KEYS) ns >>= 1;
    return ns;
}

// Can a frame of this size be binned at all?  (one sort key per 8 x 8-pixel tile at least)
bool frame_fits_binning(int W, int H)
{
    const long long tiles = (long long)((W + BIN_TILE - 1) / BIN_TILE) * ((H + BIN_TILE - 1) / BIN_TILE);
    return tiles + 64 <= (long long)BIN_MAX_KEYS;
}

// A binned frame: camera origin rows, camera-tile bins, trace.  The light-cube bins come from the shared cache when the lights
// stand still -- the only per-frame binning is then the camera's -- or, for lights that moved within the last
// LIGHT_STABLE_FRAMES frames, from this frame's own pass: their cubes (CUBE_BINS_MIN bins per side) are binned TOGETHER with the
// camera frame into the stream's pair list and expanded into the stream's rows.  Nothing of that is shared, so a moving light
// needs no barrier between the streams and no host sync (the list is sized like the camera's: from an earlier frame's count).
// The reference moves the light with keys as readily as the camera (raytracer.cpp:152-162).
constexpr int LIGHT_STABLE_FRAMES = 4;

// The cost histogram (k_prep_select): wanted when the caller asked for it or the partition is the weighted one, and then from ONE
// pass per sharded call -- the first -- so that every rank files the same sequence.  hist_prepare points the pass at the device
// words (zero between passes: k_hist_out leaves them so); hist_publish sends them to the next pinned copy of the ring, tagged
// with the sharded call they belong to, an event behind them.
bool hist_wanted() { return (g.want_hist || g.strip_rows == MIRT_PARTITION_WEIGHTED) && !(g.in_sharded && g.hist_taken); }
int hist_shift_for(int tile_rows) { int sh = 0; while (((tile_rows - 1) >> sh) + 1 > SEL_HIST_MAX) sh++; return sh; }
bool g_hist_armed = false;                       // hist_prepare armed the pass that is being enqueued
int hist_prepare(RtScratch &S, const BinFrameDesc &cam, uint64_t key, SelectOut *so)
{
    g_hist_armed = false;
    if (!hist_wanted()) return MIRT_OK;
    if (!S.d_hist) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.d_hist), sizeof(uint32_t) * SEL_HIST_MAX));
        HIP_TRY(hipMemsetAsync(S.d_hist, 0, sizeof(uint32_t) * SEL_HIST_MAX, g.stream));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.h_hist), sizeof(uint32_t) * SEL_HIST_MAX * HIST_RING, hipHostMallocDefault));
        for (int i = 0; i < HIST_RING; i++) HIP_TRY(hipEventCreateWithFlags(&S.ev_hist[i], hipEventDisableTiming));
    }
    so->hist = S.d_hist;
    so->hist_shift = hist_shift_for(cam.nbv);
    const int slot = S.hist_next;
    // (the copy about to be overwritten was filed HIST_RING passes ago; a reader only ever looks at copies whose event has fired)
    S.hist_key[slot] = 0;
    S.hist_rows[slot] = ((cam.nbv - 1) >> so->hist_shift) + 1;
    S.hist_shift[slot] = so->hist_shift;
    (void)key;
    g_hist_armed = true;
    return MIRT_OK;
}
int hist_publish(RtScratch &S)
{
    if (!g_hist_armed) return MIRT_OK;
    g_hist_armed = false;
    const int slot = S.hist_next;
    uint32_t *dst = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dst), S.h_hist + (size_t)slot * SEL_HIST_MAX, 0));
    hipLaunchKernelGGL(k_hist_out, dim3(1), dim3(SEL_HIST_MAX), 0, g.stream, S.d_hist, dst);
    HIP_TRY(hipEventRecord(S.ev_hist[slot], g.stream));
    S.hist_key[slot] = g.shard_calls + 1;        // filed under the sharded call in progress (+1: 0 means "no copy"); outside one, the calls so far
    S.hist_next = (slot + 1) % HIST_RING;
    if (g.in_sharded) g.hist_taken = true;
    return MIRT_OK;
}

// The cubes of lights that MOVE, binned by the frame itself (64 x 64 bins per face): a pass of its own in the stream's light
// scratch set L -- the lights' origin rows and per-face selection lists (k_select_faces), pairs, sort, expanded rows --, apart from
// the camera's pass, so that each is kept while only the other one's inputs change: a light key with the camera at rest
// (raytracer.cpp:152-162, 385-537) re-bins the cubes and nothing else; the camera moving under lights that have not settled into the
// shared cube yet re-bins the camera frame and nothing else.  *kept: the pass was not run.
int transient_light_pass(RtScratch &L, const RtFrame &f, const float *origins, int nlights, int cube_bins, int tshells, uint32_t per_light, uint64_t lkey, bool *kept,
                         unsigned long long *zero_hits /* nullable: the frame's hit counters, zeroed by the pass's first launch when it runs */)
{
    int rc;
    *kept = false;
    uint64_t key = 0xcbf29ce484222325ull ^ g.scene_version;
    {
        auto mix = [&](const void *p, size_t nb) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < nb; i++) { key ^= b[i]; key *= 0x100000001b3ull; } };
        mix(&lkey, 8); mix(&cube_bins, 4); mix(&tshells, 4); mix(&g.n, 4); mix(&nlights, 4);
    }
    poll_pair_count(L);
    bool fresh = !L.bin_key_valid || L.bin_key != key;
    const bool may_guess = L.last_bin_mode == nlights;
    if (!fresh && L.have_known && L.known_pairs > L.cap_used) fresh = true;   // (a kept list that turned out too small is rebuilt, so that it grows)
    static const bool reuse_off = [] { const char *e = getenv("MIRT_BIN_REUSE"); return e && atoi(e) == 0; }();
    if (!fresh && !reuse_off) { *kept = true; return MIRT_OK; }
    if ((rc = ensure_face_lists(L, nlights))) return rc;
    if (nlights > L.light_tab_lights || L.light_tab_n != g.n) {
        HIP_TRY(hipStreamSynchronize(g.stream));
        L.light_tab_lights = 0;
        if ((rc = dev_realloc(&L.d_light_tab, (size_t)nlights * g.n))) return rc;
        L.light_tab_lights = nlights;
        L.light_tab_n = g.n;
    }
    const uint32_t nkeys = per_light * (uint32_t)nlights;
    if (nkeys + 1 > L.cap_bins) {
        HIP_TRY(hipStreamSynchronize(g.stream));
        L.cap_bins = 0;
        if ((rc = dev_realloc(&L.d_bin_off, (size_t)nkeys + 1))) return rc;
        L.cap_bins = nkeys + 1;
    }
    // frame descriptors of the cubes and the origins in ONE buffer, one upload: [6 * nlights descriptors | (1 + nlights) x 3 floats]
    if (!L.d_frames) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&L.d_frames), sizeof(BinFrameDesc) * (6 * MIRT_MAX_LIGHTS) + sizeof(float) * 3 * (1 + MIRT_MAX_LIGHTS)));
    struct { BinFrameDesc frames[6 * MIRT_MAX_LIGHTS]; float origins[3 * (1 + MIRT_MAX_LIGHTS)]; } up;
    static_assert(sizeof(BinFrameDesc) % 4 == 0, "descriptors are uploaded as words");
    fill_light_frames(up.frames, f, nlights, cube_bins, tshells, 0u);
    float *d_origins = reinterpret_cast<float *>(L.d_frames + 6 * nlights);
    memcpy(reinterpret_cast<char *>(up.frames + 6 * nlights), origins, sizeof(float) * 3 * (1 + nlights));      // (right behind the descriptors in use)
    // (the same launch zeroes the pass's pair counter and the face lists' lengths)
    const ZeroJob zj = { L.d_bin_counters, (int)(LIGHT_COUNTER_BYTES / 4), reinterpret_cast<uint32_t *>(zero_hits), zero_hits ? 2 * HIT_SHARDS * HIT_SHARD_STRIDE : 0 };
    HIP_TRY(upload_small(L.d_frames, &up, sizeof(BinFrameDesc) * 6 * nlights + sizeof(float) * 3 * (1 + nlights), g.stream, &zj));
    hipLaunchKernelGGL(k_select_faces, dim3((unsigned)std::min<long long>(((long long)g.n + 1023) / 1024, (long long)g.cu_count), nlights), dim3(1024), 0, g.stream,
                       g.d_tris, g.n, d_origins, L.d_frames, L.d_light_tab, L.d_face_sel, (uint32_t)g.n, L.d_face_counts);
    BinSet bs;
    memset(&bs, 0, sizeof bs);
    bs.frames = L.d_frames; bs.nframes = 6 * nlights; bs.nbins = nkeys; bs.bin_off = L.d_bin_off;
    bs.face_lists = L.d_face_sel; bs.face_counts = L.d_face_counts; bs.face_stride = (uint32_t)g.n;
    if ((rc = bin_pass(L, bs, nullptr, L.d_light_tab, L.d_bin_counters, L.d_bin_off, true, &L.bin_entries, may_guess))) return rc;
    L.last_bin_mode = nlights;
    L.bin_key = key;
    L.bin_key_valid = true;
    if (L.cap_light_rows < L.cap_entries) {                  // one row per pair at most; grown with the pair list (rare)
        HIP_TRY(hipStreamSynchronize(g.stream));
        L.cap_light_rows = 0;
        if ((rc = dev_realloc(&L.d_light_rows, (size_t)L.cap_entries))) return rc;
        L.cap_light_rows = L.cap_entries;
    }
    const uint32_t expect = std::max<uint32_t>(L.bin_entries, 1u);
    hipLaunchKernelGGL(k_expand_light_rows, dim3((unsigned)std::min<uint32_t>((expect + 255) / 256, 4096u)), dim3(256), 0, g.stream,
                       L.d_bin_off, L.d_entries, nlights, per_light, L.d_light_tab, g.n, L.d_light_rows, L.d_bin_counters, L.cap_used, (uint32_t *)nullptr);
    return MIRT_OK;
}

int rt_enqueue_binned(RtFrame &f, const mirt_view *view, RtScratch &S, RtScratch &L, const float *origins, int nlights, int y0, int y1)
{
    int rc;
    g.stats.mode_used = MIRT_RT_BINNED;
    g.stats_sel_count = nullptr;
    // light-cube resolution: bins per face side.  Finer grids shorten the shadow lists; the shared bins are built once per
    // (scene, lights), not per frame, so what they cost is memory (48 bytes per (bin, triangle) pair) and ~1 ms of build for
    // 100 k triangles.  Measured on the 100 k soup at 1080p (round 2's trace kernel, lists not yet ordered by depth): 64: 153 us,
    // 128: 125 us, 256: 105 us.  MIRT_CUBE_BINS=64|128|256 fixes the grid (and keeps every frame on the shared cache).
    static const int cube_override = [] { const char *e = getenv("MIRT_CUBE_BINS"); return e ? atoi(e) : 0; }();
    int fine_bins = g.n < 2000 ? CUBE_BINS_MIN : (g.n < 20000 ? 2 * CUBE_BINS_MIN : 4 * CUBE_BINS_MIN);
    const bool fixed_grid = cube_override == 64 || cube_override == 128 || cube_override == 256;
    if (fixed_grid) fine_bins = cube_override;
    // (many light positions -- 16 soft-shadow samples of two lights -- at the finest grid are more keys than one sort pass holds)
    while (fine_bins > CUBE_BINS_MIN && 6ll * fine_bins * fine_bins * nlights * 4 > (long long)BIN_MAX_KEYS) fine_bins /= 2;

    const uint64_t lkey = light_key_of(origins, nlights);
    if (g.lc.track_key == lkey) g.lc.stable++;
    else { g.lc.track_key = lkey; g.lc.stable = 0; }
    const bool cached = g.lc.valid && g.lc.key == lkey && g.lc.cube_bins == fine_bins;
    const bool transient = nlights > 0 && !fixed_grid && !cached && g.lc.stable < LIGHT_STABLE_FRAMES;

    k_begin(MIRT_K_BIN);
    if (!transient && (rc = light_cache_ensure(L, f, origins, nlights, fine_bins))) return rc;   // (in the light pass's scratch: the camera's tables stay)
    const int cube_bins = transient ? CUBE_BINS_MIN : fine_bins;

    BinSet bs;
    memset(&bs, 0, sizeof bs);
    bs.frame0 = make_camera_frame(view, y0, y1, g.aa);
    bs.frames = nullptr; bs.nframes = 1;
    // The camera's sort keys are LOCAL to the rows the call renders: tile (i, j) of a band that starts at tile row j0 has bin
    // (j - j0) * nbu + i (the frame's `base` is -j0 * nbu, modulo 2^32), so a band of a sharded frame sorts an eighth of the keys
    // -- buckets an eighth as wide, spread over all the sort's workgroups -- and writes an eighth of the offsets.  (With the whole
    // frame's key space a band's pairs sat in an eighth of the buckets: k_bs_local took 94 us for a middle band of the 1 M-triangle
    // frame at 8K against 112 us for the whole frame.)  The kernels that index the offsets by the frame's tile number get the
    // array's base shifted accordingly (cam_off below).
    const int band_tile_rows = bs.frame0.j1 - bs.frame0.j0;
    const uint32_t key_shift_tiles = (uint32_t)bs.frame0.j0 * (uint32_t)bs.frame0.nbu;
    bs.frame0.base = 0u - key_shift_tiles;
    {
        // depth shells: the tiles' lists come out of the sort roughly front to back (key = bin * shells + shell of the
        // candidate's `near` bound, uniform steps between the nearest and the farthest point of the scene's box)
        const int ns = camera_shells_for((long long)bs.frame0.nbu * band_tile_rows);
        double dn = 0.0, df = 0.0;
        const bool okr = shell_range(view->pos, &dn, &df);
        bs.frame0.nshell = okr ? ns : 1;
        bs.frame0.shell_d0 = (float)dn;
        bs.frame0.shell_iw = okr ? (float)(ns / (df - dn)) : 0.0f;
    }
    const uint32_t cam_keys = (uint32_t)bs.frame0.nbu * (uint32_t)band_tile_rows * (uint32_t)bs.frame0.nshell;
    // this frame's own light cubes (moving lights) are a pass of their own, with keys of their own (below)
    const int tshells = transient ? light_shells_for(nlights, cube_bins, 0u) : 1;
    const uint32_t per_light = 6u * (uint32_t)(cube_bins * cube_bins) * (uint32_t)tshells;
    bs.nbins = cam_keys;
    if (bs.nbins + 1 > S.cap_bins) {
        const size_t cap = (size_t)bs.nbins + 1;
        HIP_TRY(hipStreamSynchronize(g.stream));             // (a frame of this stream may still read the old array)
        if ((rc = dev_realloc(&S.d_bin_off, cap))) { S.cap_bins = 0; return rc; }
        S.cap_bins = (uint32_t)cap;
        S.bin_key_valid = false;
    }
    // Zero-fill ON the stream that uses the buffer: hipMemset runs on the null stream, which the library's non-blocking streams
    // are not ordered with -- with several processes on one device (three ranks rehearsing a sharded run) such a fill has been seen
    // to land AFTER the first kernels of g.stream had started counting, which cut the pair count short (a light cube built from
    // it kept wrong shadows until the lights moved; a camera pass failed with "produced N pairs twice").
    if (!S.d_bin_counters) { HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.d_bin_counters), 512)); HIP_TRY(hipMemsetAsync(S.d_bin_counters, 0, 512, g.stream)); }
    if (S.sel_n != g.n) {
        HIP_TRY(hipStreamSynchronize(g.stream));
        S.sel_n = 0;
        if ((rc = dev_realloc(&S.d_sel, (size_t)g.n))) return rc;
        S.sel_n = g.n;
        S.bin_key_valid = false;
    }
    bs.bin_off = S.d_bin_off;

    uint64_t key = 0xcbf29ce484222325ull ^ g.scene_version;
    {
        auto mix = [&](const void *p, size_t nb) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < nb; i++) { key ^= b[i]; key *= 0x100000001b3ull; } };
        mix(view, sizeof *view); mix(&y0, 4); mix(&y1, 4); mix(&g.n, 4); mix(&g.aa, 4);
    }
    const int bin_mode = 0;                                  // (the camera's pass bins the camera frame alone)
    g.hits_clean[g.hits_cur] = false;
    if (!S.d_frames) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.d_frames), sizeof(BinFrameDesc) * (1 + 6 * MIRT_MAX_LIGHTS)));
    // The pair count is read back (4 bytes + one sync of this stream) only when the inputs that determine it changed AND no
    // count of an earlier pass of the same kind is at hand (a camera-only count says nothing about camera + light cubes).
    // When NOTHING the pass depends on has changed since this stream's last pass -- the view stands still while a light key, a
    // toggle or nothing at all asks for a frame (raytracer.cpp:385-537 set isUpdated without touching cameraPos / yaw) -- the
    // stream still holds that pass's tables: origin rows, selection, sorted lists, offsets, tile order, and with them the rows
    // of lights binned by the frame.  The frame then starts at the trace kernel.  (A kept pass whose list turned out too small
    // -- its published count says so -- is redone, so that the list grows: bin_pass.)
    poll_pair_count(S);
    bool fresh = !S.bin_key_valid || S.bin_key != key;
    const bool may_guess = S.last_bin_mode == bin_mode;
    if (!fresh && S.have_known && S.known_pairs > S.cap_used) fresh = true;
    static const bool reuse_off = [] { const char *e = getenv("MIRT_BIN_REUSE"); return e && atoi(e) == 0; }();
    const bool reuse = !fresh && !reuse_off;
    const uint32_t pairs_x = (uint32_t)((bs.frame0.nbu + 1) / 2);
    uint32_t group_rows[ORDER_GROUPS] = { 0 };
    for (int j = bs.frame0.j0; j < bs.frame0.j1; j++) group_rows[((uint32_t)j >> ORDER_STRIPE_SHIFT) & (ORDER_GROUPS - 1)]++;
    const uint32_t order_seg = pairs_x * *std::max_element(group_rows, group_rows + ORDER_GROUPS);
    const uint32_t *cam_off = S.d_bin_off - (size_t)key_shift_tiles * (size_t)bs.frame0.nshell;   // indexed by the FRAME's tile number
    // The cubes of lights that moved within the last frames: a pass of their own (transient_light_pass), kept while the lights stand
    // still.  When the camera's pass runs as well, the light pass goes FIRST and onto the stream's side stream: the two chains
    // share nothing until the trace kernel and are each bound by the latency of their launches, so side by side they take the longer
    // one's time instead of the sum (one frame in flight, camera and light moving: 0.268 ms one after the other, see
    // profiles/r04_moving_light.txt for the figure side by side).  The side stream starts behind everything the main stream has
    // queued (the previous frame's trace kernel reads the tables the pass rewrites) and is joined in front of this frame's.
    bool lights_kept = false, forked = false;
    if (transient) {
        static const bool side_off = [] { const char *e = getenv("MIRT_LIGHT_SIDE_STREAM"); return e && atoi(e) == 0; }();
        hipStream_t main_stream = g.stream;
        forked = !reuse && !side_off;
        if (forked) {
            HIP_TRY(hipEventRecord(g.ev_fork[g.si], main_stream));
            HIP_TRY(hipStreamWaitEvent(g.aux[g.si], g.ev_fork[g.si], 0));
            g.stream = g.aux[g.si];
        }
        // (with the camera's pass kept nothing else runs in front of the trace kernel: the light pass's first launch zeroes the hit counters too)
        rc = transient_light_pass(L, f, origins, nlights, cube_bins, tshells, per_light, lkey, &lights_kept, reuse ? g.d_hits : nullptr);
        g.stream = main_stream;
        if (rc) return rc;
        if (forked) HIP_TRY(hipEventRecord(g.ev_join[g.si], g.aux[g.si]));
    }
    if (reuse) {
        // (the first kernel of a pass zeroes the frame's hit counters on the way; here nothing runs in front of the trace kernel --
        // unless the light pass has just run and done it)
        if (!(transient && !lights_kept)) HIP_TRY(hipMemsetAsync(g.d_hits, 0, sizeof(unsigned long long) * HIT_SHARDS * HIT_SHARD_STRIDE, g.stream));
        g.stats.bins_reused = 1;
    } else {
        // first kernel of the frame: the camera's origin rows for the triangles the rows of this call can see, and their list
        // (k_prep_select); it also zeroes the hit counters and the pass's counters
        S.sel_parity ^= 1;
        SelectOut so;
        memset(&so, 0, sizeof so);
        so.cam_tab = S.d_cam_tab; so.sel = S.d_sel;
        so.sel_count = S.d_bin_counters + SEL_COUNT0 + S.sel_parity; so.sel_count_next = S.d_bin_counters + SEL_COUNT0 + (S.sel_parity ^ 1);
        so.zero_hits = g.d_hits; so.zero_counter = S.d_bin_counters;
        if ((rc = hist_prepare(S, bs.frame0, key, &so))) return rc;
        // one workgroup of 1024 threads per CU: a workgroup reserves its slice of the list with ONE atomic (rt_binned.hip)
        const unsigned sel_grid = (unsigned)std::min<long long>(((long long)g.n + 1023) / 1024, (long long)g.cu_count);
        hipLaunchKernelGGL(k_prep_select, dim3(sel_grid), dim3(1024), 0, g.stream, g.d_tris, g.n, bs.frame0, so);
        if ((rc = hist_publish(S))) return rc;
        bs.sel = S.d_sel; bs.sel_count = so.sel_count;
        g.stats_sel_count = so.sel_count;
        if ((rc = bin_pass(S, bs, S.d_cam_tab, nullptr, S.d_bin_counters, S.d_bin_off, true, &S.bin_entries, may_guess))) return rc;
        S.last_bin_mode = bin_mode;
        S.bin_key = key;
        S.bin_key_valid = true;
        // the order the trace kernel's waves take the tile pairs in: per XCD group (pairs of tile rows dealt round-robin), longest
        // lists first
        if ((size_t)order_seg > S.cap_order) {
            HIP_TRY(hipStreamSynchronize(g.stream));
            S.cap_order = 0;
            if ((rc = dev_realloc(&S.d_order, (size_t)ORDER_GROUPS * ORDER_CLASSES * order_seg))) return rc;
            S.cap_order = order_seg;
        }
        hipLaunchKernelGGL(k_tile_order, dim3((pairs_x + 63) / 64, (unsigned)(bs.frame0.j1 - bs.frame0.j0)), dim3(64), 0, g.stream, cam_off, bs.frame0.nshell,
                           bs.frame0.nbu, bs.frame0.j0, bs.frame0.j1, S.d_bin_counters, S.cap_used, S.d_order, order_seg);
    }
    if (forked) HIP_TRY(hipStreamWaitEvent(g.stream, g.ev_join[g.si], 0));
    (void)lights_kept;
    k_end(MIRT_K_BIN);

    RtTraceFrame tf;
    memset(&tf, 0, sizeof tf);
    tf.f = f;
    tf.f.cam_tab = S.d_cam_tab;
    // (a frame whose pair list overflowed walks the origin tables themselves: every triangle for every ray)
    tf.f.light_tab = transient ? L.d_light_tab : g.lc.d_light_tab;
    tf.f.unsafe = nullptr;
    tf.cam_off = cam_off;
    tf.cam_entries = S.d_entries;
    tf.sel = S.d_sel; tf.sel_count = S.d_bin_counters + SEL_COUNT0 + S.sel_parity;
    // geometry rows staged with every candidate while the scene's tables fit the caches, fetched by the exact stage beyond (rt_trace.hip);
    // MIRT_LAZY_GEO=0|1 fixes the choice
    static const int lazy_env = [] { const char *e = getenv("MIRT_LAZY_GEO"); return e ? atoi(e) : -1; }();
    tf.lazy_geo = lazy_env >= 0 ? (lazy_env != 0) : (g.n >= 400000);
    tf.geo = g.d_geo;
    tf.shade = g.d_shade;
    tf.light_off = transient ? L.d_bin_off : g.lc.d_off;
    tf.light_rows = transient ? L.d_light_rows : g.lc.d_rows;
    tf.light_tri = transient ? L.d_entries : g.lc.d_row_tri;
    tf.light_frames = transient ? L.d_frames : g.lc.d_frames;
    tf.tiles_x = bs.frame0.nbu;
    tf.cube_bins = cube_bins;
    tf.cam_shells = bs.frame0.nshell;
    tf.light_shells = transient ? tshells : g.lc.shells;
    tf.pair_count = S.d_bin_counters;
    tf.pair_cap = S.cap_used;
    // (lights binned by the frame: their own pass's count; the shared cube's tables are complete by construction)
    tf.light_pair_count = transient ? L.d_bin_counters : nullptr;
    tf.light_pair_cap = L.cap_used;
    // one wave per pair of 8 x 8 tiles
    tf.order = S.d_order; tf.order_count = S.d_bin_counters + 16; tf.order_seg = order_seg;
    // (waves never synchronise with each other: one-wave workgroups are the finest scheduling unit; 84 / 87 / 89 us with 1 / 2 / 4)
    const dim3 tgrid(ORDER_GROUPS * order_seg);             // (workgroup id % 8 = XCD group, id / 8 = the wave among the group's)
    const size_t lds = rt_trace_lds_bytes(1);
    k_begin(MIRT_K_TRACE);
    // (the kernel's own statistics -- tests, candidates, steps, drains -- only for frames rendered with profiling on: rt_trace.hip)
    g.pending_counted = g.profiling;
    if (f.aa > 1) {
        if (g.profiling) hipLaunchKernelGGL((k_rt_trace2<true, true>), tgrid, dim3(64), lds, g.stream, tf);
        else hipLaunchKernelGGL((k_rt_trace2<true, false>), tgrid, dim3(64), lds, g.stream, tf);
    } else {
        // five waves per SIMD (the 96-VGPR instantiation) for the large scenes and for the frame that runs alone, four (98 VGPRs) for
        // the small scenes' frames in flight: rt_trace.hip says why; MIRT_TR_WAVES5=0|1 fixes the choice
        static const int waves5_env = [] { const char *e = getenv("MIRT_TR_WAVES5"); return e ? atoi(e) : -1; }();
        const bool waves5 = waves5_env >= 0 ? (waves5_env != 0) : (g.n >= 400000 || g.in_flight <= 2);
        if (g.profiling) hipLaunchKernelGGL((k_rt_trace2<false, true>), tgrid, dim3(64), lds, g.stream, tf);
        else if (waves5) hipLaunchKernelGGL((k_rt_trace2<false, false, 5>), tgrid, dim3(64), lds, g.stream, tf);
        else hipLaunchKernelGGL((k_rt_trace2<false, false>), tgrid, dim3(64), lds, g.stream, tf);
    }
    k_end(MIRT_K_TRACE);
    HIP_TRY(hipGetLastError());
    for (RtScratch *P : { &S, &L })
        if (P->count_event_due) {
            P->count_event_due = false;
            HIP_TRY(hipEventRecord(P->ev_count, g.stream));
            P->count_pending = true;
        }
    call_end();
    return MIRT_OK;
}

int rt_enqueue(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect, int mode,
               int y0, int y1, int row_origin, void *d_xrgb, int pitch_bytes, void *d_rgb, void *d_index, void *d_fd = nullptr,
               void *d_dist = nullptr, void *d_pos = nullptr)
{
    int rc;
    if ((rc = need_init())) return rc;
    if ((rc = check_view(view, lights, nlights, indirect))) return rc;
    if (g.n <= 0) return fail(MIRT_ERR_NO_SCENE, "no scene uploaded (mirt_scene_upload)");
    if (!d_xrgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "xrgb output must not be NULL");
    if (y0 < 0 || y1 > view->height || y0 > y1) return fail(MIRT_ERR_INVALID_ARGUMENT, "row band [%d,%d) outside [0,%d)", y0, y1, view->height);
    if (pitch_bytes < view->width * 4 || (pitch_bytes & 3)) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, view->width);
    if (mode != MIRT_RT_AUTO && mode != MIRT_RT_BRUTE && mode != MIRT_RT_BINNED) return fail(MIRT_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);

    const int light_positions = nlights * (g.soft_samples > 1 ? g.soft_samples : 1);    // shadow-ray origins
    if (light_positions > MIRT_MAX_LIGHTS)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "%d lights x %d soft-shadow samples exceed %d light positions", nlights, g.soft_samples, MIRT_MAX_LIGHTS);

    RtFrame f;
    memset(&f, 0, sizeof f);
    f.tris15 = g.d_tris;
    f.n = g.n;
    memcpy(f.cam, view->pos, sizeof f.cam);
    memcpy(f.rot, view->rot, sizeof f.rot);
    f.focal = view->focal;
    f.W = view->width;
    f.H = view->height;
    // Light positions the shadow rays start from: the lights themselves, or with soft shadows `samples` jittered
    // positions per light (randomPositions[k*SOFT_SHADOWS_SAMPLES + i], raytracer.cpp:286), each carrying the
    // light's colour*intensity (:282) which light_term() divides by samples (:296).
    const int samples = g.soft_samples > 1 ? g.soft_samples : 1;
    const int npos = nlights * samples;
    if (npos > MIRT_MAX_LIGHTS) return fail(MIRT_ERR_INVALID_ARGUMENT, "%d lights x %d soft-shadow samples exceed %d light positions", nlights, samples, MIRT_MAX_LIGHTS);
    if (samples > 1 && npos > g.soft_npos) return fail(MIRT_ERR_INVALID_ARGUMENT, "%d jittered positions needed, %d were set (mirt_set_soft_shadows)", npos, g.soft_npos);
    f.nlights = npos;
    f.samples = samples;
    f.aa = g.aa > 1 ? g.aa : 1;
    float origins[(1 + MIRT_MAX_LIGHTS) * 3];
    memcpy(origins, view->pos, 12);
    for (int j = 0; j < npos; j++) {
        const int k = j / samples;
        const float *pos = samples > 1 ? g.soft_pos + 3 * j : lights[k].pos;
        memcpy(f.lpos[j], pos, 12);
        memcpy(origins + 3 * (j + 1), pos, 12);
        // P = (color * intensity) / samples (raytracer.cpp:282, :296): uniform per light, so the division happens once here
        // (host float division is the same IEEE operation the kernels would run per pixel)
        for (int c = 0; c < 3; c++) f.lcol[j][c] = (lights[k].color[c] * lights[k].intensity) / (float)f.samples;
    }
    f.lights_in_range = 1;
    for (int j = 0; j < npos; j++) f.lights_in_range &= light_colour_in_range(f.lcol[j]) ? 1 : 0;
    nlights = npos;            // from here on "lights" means light positions
    memcpy(f.indirect, indirect, 12);
    f.y0 = y0; f.y1 = y1; f.row_origin = row_origin;
    f.xrgb = static_cast<uint32_t *>(d_xrgb);
    f.pitch_words = pitch_bytes / 4;
    f.rgb = static_cast<float *>(d_rgb);
    f.index = static_cast<int32_t *>(d_index);
    f.fd = static_cast<float *>(d_fd);
    f.dist = static_cast<float *>(d_dist);
    f.pos = static_cast<float *>(d_pos);
    f.focal_plane = g.dof_focal;
    // The pre-reject filter is proven for finite, moderate operands only (rt_common.hpp); anything else
    // (absurd coordinates, NaN/Inf) renders through the exact-only path.  Ray directions of the primary
    // rays are bounded by 3 * max|rot| * max(W, H, |focal|).
    float rmax = 0.0f;
    for (int i = 0; i < 9; i++) rmax = fmaxf(rmax, fabsf(view->rot[i]));
    const float dmax = 3.0f * rmax * fmaxf(fmaxf((float)view->width, (float)view->height), fabsf(view->focal));
    bool safe = g.scene_finite && finite_below(view->rot, 9, 1.0e6f) && (dmax < 1.0e6f) &&
                finite_below(origins, 3 * (1 + nlights), 1.0e8f);      // camera and light positions
    const uint32_t flags_init[4] = { safe ? 0u : 1u, 0u, 0u, 0u };

    // ---- mode: brute force for small scenes, binned otherwise; unsafe operands always render exact brute ----
    // MIRT_RT_AUTO bins when the scene is beyond the tile kernel (65 triangles or more) and the brute-force work, pixels x
    // triangles, is above ~4e7: binning + sorting costs ~40 us whatever the scene, brute force ~7.5e-10 ms per pixel-triangle
    // (tools/threshold_sweep.py at 1080p: 65 triangles 0.099 vs 0.043 ms, 300: 0.47 vs 0.079, 800: 1.13 vs 0.097).
    static const int auto_threshold = [] { const char *e = getenv("MIRT_BIN_THRESHOLD"); return e ? atoi(e) : 65; }();
    bool binned = (mode == MIRT_RT_BINNED) ||
                  (mode == MIRT_RT_AUTO && g.n >= auto_threshold && (long long)view->width * (y1 - y0) > 4096 &&
                   (long long)view->width * (y1 - y0) * g.n >= 40000000LL);
    if (!safe) binned = false;
    if (binned && !frame_fits_binning(view->width, view->height)) {
        // more 8 x 8-pixel tiles than one sort pass has keys (a frame beyond ~23 000 x 23 000 pixels)
        if (mode == MIRT_RT_BINNED) return fail(MIRT_ERR_INVALID_ARGUMENT, "frame %dx%d has more tiles than the binned path can key; use MIRT_RT_AUTO or row bands of a smaller frame", view->width, view->height);
        binned = false;
    }
    const int rows = y1 - y0;

    // Small scenes (the reference's own 30-triangle Cornell box): one launch, every table built in LDS by the
    // workgroup itself -- no origin-table kernel, no global loads inside the loops.
    // Scenes of at most 64 triangles (the reference's Cornell box has 30): per-tile candidate masks, one lane per
    // triangle (rt_tile.hip).  Needs operands inside the filter's proven range, like binning does.
    const size_t tile_lds = (size_t)g.n * 16 * (12 + 3 * nlights);
    const bool tile_path = !binned && safe && g.n <= 64 && tile_lds <= 64 * 1024;

    // A frame reads the scene and writes the caller's planes plus its stream's own tables, counters and depth-of-field
    // planes, so frames may overlap (call_begin).
    call_begin();
    g.pending_is_rt = true;
    g.pending_primary = (uint64_t)view->width * (uint64_t)(y1 - y0) * (uint64_t)((g.aa > 1 ? g.aa : 1) * (g.aa > 1 ? g.aa : 1));
    g.pending_nlights = light_positions;
    g.stats.mode_used = MIRT_RT_BRUTE;
    g.pending_empty = (y1 == y0);
    g.pending_counted = false;
    if (y1 == y0) { call_end(); return MIRT_OK; }
    // hit counters: every stream owns two buffers used alternately, so that a kernel can clear the one the NEXT frame
    // on its stream will use
    const int si = g.si;
    g.hits_tog[si] ^= 1;
    g.hits_cur = si + MAX_FLIGHT * g.hits_tog[si];
    g.d_hits = g.d_hits2[g.hits_cur];
    f.hit_count = g.d_hits;
    RtScratch &S = g.rt[si];
    if (!tile_path) {                            // origin tables of this stream, sized for the scene and the light positions
        if (S.cam_tab_n != g.n) {
            S.cam_tab_n = 0;
            if ((rc = dev_realloc(&S.d_cam_tab, (size_t)g.n))) return rc;
            S.cam_tab_n = g.n;
        }
        if (!binned && (light_positions > S.light_tab_lights || S.light_tab_n != g.n)) {   // (binned frames read the shared light cache)
            S.light_tab_lights = 0;
            if ((rc = dev_realloc(&S.d_light_tab, (size_t)light_positions * g.n))) return rc;
            S.light_tab_lights = light_positions;
            S.light_tab_n = g.n;
        }
        if (!S.d_origins) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.d_origins), sizeof(float) * 3 * (1 + MIRT_MAX_LIGHTS)));
        if (!S.d_flags) { HIP_TRY(hipMalloc(reinterpret_cast<void **>(&S.d_flags), 16)); HIP_TRY(hipMemsetAsync(S.d_flags, 0, 16, g.stream)); }
    }
    f.cam_tab = S.d_cam_tab;
    f.light_tab = S.d_light_tab;
    f.unsafe = S.d_flags;

    if (tile_path) {
        RtTileFrame tf;
        memset(&tf, 0, sizeof tf);
        tf.f = f;
        tf.cam = make_camera_frame(view, y0, y1, g.aa);
        // a wave owns a 16 x 8-pixel tile, two pixels per lane (packed FP32, rt_tile.hip); workgroups of 4 waves, one workgroup
        // per 4 tiles.  Tile cost varies several-fold (candidates, shadowed or lit), and the hardware's dynamic workgroup dispatch
        // balances that better than any static assignment (measured on the Cornell box at 1080p: 40.5 us with one tile per wave,
        // 46 us with a resident grid striding over the tiles, 49 us with 3 tiles per wave).
        const int tw = 16, th = 8, wpb = 4;
        tf.tiles_x = (view->width + tw - 1) / tw;
        tf.tiles_y = (rows + th - 1) / th;
        const long long ntiles = (long long)tf.tiles_x * tf.tiles_y;
        const unsigned blocks = (unsigned)((ntiles + wpb - 1) / wpb);
        if (!g.hits_clean[g.hits_cur])
            HIP_TRY(hipMemsetAsync(g.d_hits, 0, sizeof(unsigned long long) * HIT_SHARDS * HIT_SHARD_STRIDE, g.stream));
        g.hits_clean[g.hits_cur] = false;
        g.pending_counted = true;
        const int hits_other = si + MAX_FLIGHT * (g.hits_tog[si] ^ 1);
        tf.clear_hits = g.d_hits2[hits_other];           // zeroed by this launch for the next frame on this stream: no memset node per frame
        g.hits_clean[hits_other] = true;
        // Tables: built once per frame by k_tile_tables when the frame has enough workgroups to make rebuilding them in
        // each one the larger cost; small frames are bound by the launch rate and keep the single launch.
        tf.tables = blocks >= 1024u ? g.d_tile_tab[si] : nullptr;   // one table buffer per stream
        if (tf.tables) {
            k_begin(MIRT_K_PREP);
            hipLaunchKernelGGL(k_tile_tables, dim3(1), dim3(64), 0, g.stream, tf);
            k_end(MIRT_K_PREP);
        }
        k_begin(MIRT_K_TRACE);
        if (f.aa > 1) hipLaunchKernelGGL((k_rt_tile2<16, true>), dim3(blocks), dim3(64 * wpb), tile_lds, g.stream, tf);
        else hipLaunchKernelGGL((k_rt_tile2<16, false>), dim3(blocks), dim3(64 * wpb), tile_lds, g.stream, tf);
        k_end(MIRT_K_TRACE);
        HIP_TRY(hipGetLastError());
        call_end();
        return MIRT_OK;
    }

    const size_t small_lds = 16 + (size_t)g.n * sizeof(OriginRow) * (2 + nlights);
    if (!binned && small_lds <= 48 * 1024) {
        HIP_TRY(hipMemsetAsync(g.d_hits, 0, sizeof(unsigned long long) * HIT_SHARDS * HIT_SHARD_STRIDE, g.stream));
        g.hits_clean[g.hits_cur] = false;
        k_begin(MIRT_K_TRACE);
        // (two rays per lane, packed FP32 filter: 188 -> 163 ms on the 100 k soup against one)
        hipLaunchKernelGGL(k_rt_small<2>, dim3((view->width + 127) / 128, (rows + 3) / 4), dim3(256), small_lds, g.stream, f, safe ? 0 : 1);
        k_end(MIRT_K_TRACE);
        HIP_TRY(hipGetLastError());
        call_end();
        return MIRT_OK;
    }

    if (binned) return rt_enqueue_binned(f, view, S, g.rt_lt[si], origins, nlights, y0, y1);

    HIP_TRY(upload_small(S.d_flags, flags_init, sizeof flags_init, g.stream));
    S.bin_key_valid = false;                     // (k_prep_origin below overwrites the camera rows a kept binning pass would count on)
    g.hits_clean[g.hits_cur] = false;
    HIP_TRY(upload_small(S.d_origins, origins, sizeof(float) * 3 * (1 + nlights), g.stream));

    k_begin(MIRT_K_PREP);
    hipLaunchKernelGGL(k_prep_origin, dim3((g.n + 255) / 256, 1 + nlights), dim3(256), 0, g.stream,
                       g.d_tris, g.n, S.d_origins, V3(0.0f, 0.0f, 0.0f), 0, S.d_cam_tab, S.d_light_tab, S.d_flags, g.d_hits, (uint32_t *)nullptr);
    k_end(MIRT_K_PREP);

    if (g.aa <= 1 && (long long)view->width * rows <= 4096 && g.n >= 1024) {
        // few rays, many triangles: one wave per ray, lanes over triangles, wavefront min-t reduce
        const long long nrays = (long long)view->width * rows;
        k_begin(MIRT_K_TRACE);
        hipLaunchKernelGGL(k_rt_wave, dim3((unsigned)((nrays + 3) / 4)), dim3(256), 0, g.stream, f);
        k_end(MIRT_K_TRACE);
        HIP_TRY(hipGetLastError());
        call_end();
        return MIRT_OK;
    }
    const size_t lds = (size_t)(g.n < RT_CHUNK_ROWS ? g.n : RT_CHUNK_ROWS) * sizeof(OriginRow);
    k_begin(MIRT_K_TRACE);
    hipLaunchKernelGGL(k_rt_brute<2>, dim3((view->width + 127) / 128, (rows + 3) / 4), dim3(256), lds, g.stream, f);
    k_end(MIRT_K_TRACE);
    HIP_TRY(hipGetLastError());
    call_end();
    return MIRT_OK;
}

// The device alias of a host pointer inside a registered surface (rows [0, H) of `pitch` bytes must fit), or NULL.
char *registered_alias(const void *host, size_t pitch, int H)
{
    const char *p = static_cast<const char *>(host);
    for (const Ctx::HostSurface &r : g.surf)
        if (r.host && p >= r.host && p + pitch * (size_t)H <= r.host + r.bytes) return r.dev + (p - r.host);
    return nullptr;
}

// How a frame reaches a REGISTERED host surface: 0 (default) = device staging plane + one DMA copy into the pinned surface,
// 1 = the render kernels store their XRGB words straight into the mapped surface (no staging plane, no copy; the stores cross
// the link while the frame is still being computed).  MIRT_HOST_PATH=direct|dma.  Measured on the MI355X box (bench.py
// host_path): 1080p ray tracer 0.226 (dma) / 0.230 (direct) / 0.224 ms (unregistered, pageable) per frame, 4K rasteriser
// 0.71 (pageable) / 0.90 ms (direct) -- the runtime's own staging of pageable copies already runs at the rate the link gives
// here (37-46 GB/s), so registering buys nothing on this machine and direct stores lose to the DMA engine on large frames.
bool host_direct()
{
    static const bool direct = [] { const char *e = getenv("MIRT_HOST_PATH"); return e && !strcmp(e, "direct"); }();
    return direct;
}

int copy_plane_interior(void *dst, int dst_pitch, const void *src, int src_pitch, int W, int H)
{
    // rows 1..H-2, columns 1..W-2 only: the reference never writes the 1-pixel border (raytracer.cpp:618-620)
    if (W < 3 || H < 3) return MIRT_OK;
    HIP_TRY(hipMemcpy2DAsync(static_cast<char *>(dst) + dst_pitch + 4, dst_pitch,
                             static_cast<const char *>(src) + src_pitch + 4, src_pitch,
                             (size_t)(W - 2) * 4, H - 2, hipMemcpyDeviceToHost, g.stream));
    return MIRT_OK;
}

// Depth of field (CalculateDOF with DOF_ENABLED, raytracer.cpp:613-640 / rasteriser.cpp:494-513): the render kernels
// write pixelColours + focalDistances for the band AND the rows its blur taps reach into library-owned planes, then
// k_dof resolves the band into the caller's surface.  `render(ry0, ry1, xrgb, rgb, fd, index, zinv)` runs the path.
template <class Render>
int render_with_dof(const mirt_view *view, int y0, int y1, int row_origin, void *d_xrgb, int pitch_bytes,
                    void *user_rgb, void *user_index, void *user_zinv, bool clear_border, Render render)
{
    int rc;
    // the caller's surface reaches the blur kernel directly: validate it here, before anything is allocated or launched
    // (rt_enqueue / raster_enqueue only see the library-owned planes)
    if ((rc = need_init())) return rc;
    if (!view) return fail(MIRT_ERR_INVALID_ARGUMENT, "view must not be NULL");
    if (view->width < 1 || view->height < 1 || view->width > 32768 || view->height > 32768)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "frame size %dx%d out of range [1,32768]", view->width, view->height);
    if (y0 < 0 || y1 > view->height || y0 > y1) return fail(MIRT_ERR_INVALID_ARGUMENT, "row band [%d,%d) outside [0,%d)", y0, y1, view->height);
    if (!d_xrgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "xrgb output must not be NULL");
    if (pitch_bytes < view->width * 4 || (pitch_bytes & 3)) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, view->width);
    const int W = view->width, H = view->height, K = g.dof_k;
    const int zlo = (int)std::ceil((float)K / -2.0f), zhi = (int)std::ceil((float)K / 2.0f);
    const int reach = std::max(-zlo, zhi - 1) + 1;           // +1: a tap column outside the row wraps into the next row
    const int ry0 = std::max(0, y0 - reach), ry1 = std::min(H, y1 + reach);
    const size_t npx = (size_t)W * (size_t)(ry1 - ry0);
    // the stream call_begin() will give this frame (it is self-contained: its planes are this stream's own)
    Ctx::DofPlanes &D = g.dof[next_si()];
    if (npx > D.cap_px) {
        for (void **p : { (void **)&D.rgb, (void **)&D.fd, (void **)&D.xrgb, (void **)&D.index, (void **)&D.zinv }) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        D.cap_px = 0;
        if (hipMalloc((void **)&D.rgb, npx * 12) != hipSuccess || hipMalloc((void **)&D.fd, npx * 4) != hipSuccess ||
            hipMalloc((void **)&D.xrgb, npx * 4) != hipSuccess || hipMalloc((void **)&D.index, npx * 4) != hipSuccess ||
            hipMalloc((void **)&D.zinv, npx * 4) != hipSuccess)
            return fail(MIRT_ERR_OUT_OF_MEMORY, "depth-of-field planes (%zu pixels)", npx);
        D.cap_px = npx;
    }
    // the kernels index their planes with full-frame pixel numbers: shift the bases so that row ry0 is the first stored
    const ptrdiff_t shift = (ptrdiff_t)ry0 * W;
    float *rgb = D.rgb - 3 * shift, *fd = D.fd - shift, *zinv = D.zinv - shift;
    int32_t *index = D.index - shift;
    if ((rc = render(ry0, ry1, (void *)D.xrgb, (void *)rgb, (void *)fd, user_index ? (void *)index : nullptr,
                     user_zinv ? (void *)zinv : nullptr))) return rc;
    if (y1 > y0) {
        DofFrame d;
        d.rgb = rgb; d.fd = fd; d.W = W; d.H = H; d.K = K;
        d.y0 = y0; d.y1 = y1; d.row_origin = row_origin; d.ry0 = ry0; d.ry1 = ry1;
        d.xrgb = static_cast<uint32_t *>(d_xrgb); d.pitch_words = pitch_bytes / 4; d.clear_border = clear_border ? 1 : 0;
        if (g.profiling) { (void)hipEventRecord(g.ev[EV_K0 + 2 * MIRT_K_DOF], g.stream); g.ev_used[MIRT_K_DOF] = true; }
        launch_dof(d, g.stream);
        if (g.profiling) (void)hipEventRecord(g.ev[EV_K0 + 2 * MIRT_K_DOF + 1], g.stream);
        HIP_TRY(hipGetLastError());
        const size_t rows = (size_t)(y1 - y0), off = (size_t)(y0 - ry0) * W, uoff = (size_t)y0 * W;
        if (user_rgb) HIP_TRY(hipMemcpyAsync((float *)user_rgb + 3 * uoff, D.rgb + 3 * off, rows * W * 12, hipMemcpyDeviceToDevice, g.stream));
        if (user_index) HIP_TRY(hipMemcpyAsync((int32_t *)user_index + uoff, D.index + off, rows * W * 4, hipMemcpyDeviceToDevice, g.stream));
        if (user_zinv) HIP_TRY(hipMemcpyAsync((float *)user_zinv + uoff, D.zinv + off, rows * W * 4, hipMemcpyDeviceToDevice, g.stream));
        if (g.call_timed) (void)hipEventRecord(g.ev[EV_CALL1], g.stream);   // the call ends after the blur
    }
    return MIRT_OK;
}

}  // namespace
}  // namespace mirt

using namespace mirt;

// ---- lifetime ----------------------------------------------------------------------------------------

extern "C" int mirt_abi_version(void) { return MIRT_ABI_VERSION; }
extern "C" const char *mirt_last_error(void) { return g_err; }

extern "C" int mirt_init(int device)
{
    if (g.init) {
        if (g.device == device) return MIRT_OK;
        mirt_shutdown();
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MIRT_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(MIRT_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MIRT_ERR_NO_DEVICE, "device %d is %s; the kernels in this library are built for gfx950 (MI355X) only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    g.cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    for (int i = 0; i < MAX_FLIGHT; i++) {
        HIP_TRY(hipStreamCreateWithFlags(&g.streams[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_order[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_cull_read[i], hipEventDisableTiming));
        g.cull_read_src[i] = 0u;
        HIP_TRY(hipStreamCreateWithFlags(&g.aux[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_fork[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_join[i], hipEventDisableTiming));
    }
    g.stream = g.streams[0];
    g.in_flight = 1;
    g.frame_no = 0;
    g.si = 0;
    for (int si = 0; si < MAX_FLIGHT; si++) for (int i = 0; i < EV_COUNT; i++) HIP_TRY(hipEventCreate(&g.ev_sets[si][i]));
    g.ev_cur = 0; g.ev = g.ev_sets[0]; g.ev_used = g.ev_used_sets[0];
    for (int i = 0; i < 2 * MAX_FLIGHT; i++) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_hits2[i]), sizeof(unsigned long long) * HIT_SHARDS * HIT_SHARD_STRIDE));
        HIP_TRY(hipMemset(g.d_hits2[i], 0, sizeof(unsigned long long) * HIT_SHARDS * HIT_SHARD_STRIDE));   // (mirt_init ends with a device sync)
        g.hits_clean[i] = true;
    }
    for (int i = 0; i < MAX_FLIGHT; i++) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.d_tile_tab[i]), sizeof(float4) * 64 * (12 + 3 * MIRT_MAX_LIGHTS)));
    g.d_hits = g.d_hits2[0];
    g.device = device;
    HIP_TRY(hipDeviceSynchronize());             // the null-stream fills above have landed before any stream of ours runs
    g.init = true;
    return MIRT_OK;
}

extern "C" void mirt_shutdown(void)
{
    if (!g.init) return;
    (void)hipSetDevice(g.device);
    for (int i = 0; i < MAX_FLIGHT; i++) if (g.streams[i]) (void)hipStreamSynchronize(g.streams[i]);
    for (int i = 0; i < MAX_FLIGHT; i++) if (g.aux[i]) (void)hipStreamSynchronize(g.aux[i]);
    for (RtScratch *set : { g.rt, g.rt_lt })
        for (int i = 0; i < MAX_FLIGHT; i++) {
            RtScratch &S = set[i];
            for (void *p : { (void *)S.d_cam_tab, (void *)S.d_light_tab, (void *)S.d_origins, (void *)S.d_flags, (void *)S.d_frames, (void *)S.d_light_rows, (void *)S.d_order, (void *)S.d_bin_off,
                             (void *)S.d_bin_counters, (void *)S.d_entries, (void *)S.d_pair_keys, (void *)S.d_pair_vals, (void *)S.d_sorted_keys, (void *)S.d_tmp_vals, (void *)S.d_bucket,
                             (void *)S.d_sel, (void *)S.d_face_sel, (void *)S.d_hist })     // (d_face_counts lies inside d_bin_counters' block)
                if (p) (void)hipFree(p);
            if (S.h_count) (void)hipHostFree(S.h_count);
            if (S.ev_count) (void)hipEventDestroy(S.ev_count);
            if (S.h_hist) (void)hipHostFree(S.h_hist);
            for (hipEvent_t e : S.ev_hist) if (e) (void)hipEventDestroy(e);
        }
    for (void *p : { (void *)g.d_geo, (void *)g.d_shade, (void *)g.lc.d_light_tab, (void *)g.lc.d_frames, (void *)g.lc.d_off, (void *)g.lc.d_rows, (void *)g.lc.d_row_tri, (void *)g.lc.d_origins, (void *)g.lc.d_counter })
        if (p) (void)hipFree(p);
    for (void *p : { (void *)g.d_tris, (void *)g.d_culled, g.d_xrgb, g.d_rgb, g.d_index, g.d_zinv, g.d_pos })
        if (p) (void)hipFree(p);
    for (unsigned long long *p : g.d_hits2) if (p) (void)hipFree(p);
    for (float4 *p : g.d_tile_tab) if (p) (void)hipFree(p);
    for (void *p : g.d_async) if (p) (void)hipFree(p);
    for (Ctx::DofPlanes &D : g.dof)
        for (void *p : { (void *)D.rgb, (void *)D.fd, (void *)D.xrgb, (void *)D.index, (void *)D.zinv }) if (p) (void)hipFree(p);
    for (Ctx::HostSurface &r : g.surf) if (r.host) (void)hipHostUnregister(r.host);
    if (g.comm_stream) (void)hipStreamSynchronize(g.comm_stream);
    comm_destroy(g.comm);
    for (int i = 0; i < 2; i++) { if (g.d_band[i]) (void)hipFree(g.d_band[i]); if (g.ev_sent[i]) (void)hipEventDestroy(g.ev_sent[i]); }
    if (g.ev_rendered) (void)hipEventDestroy(g.ev_rendered);
    if (g.comm_stream) (void)hipStreamDestroy(g.comm_stream);
    for (RasterScratch &R : g.raster) raster_scratch_free(R);
    for (int si = 0; si < MAX_FLIGHT; si++) for (int i = 0; i < EV_COUNT; i++) if (g.ev_sets[si][i]) { (void)hipEventDestroy(g.ev_sets[si][i]); g.ev_sets[si][i] = nullptr; }
    for (hipEvent_t e : g.ev_order) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_cull_read) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_fork) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g.ev_join) if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < MAX_FLIGHT; i++) if (g.aux[i]) (void)hipStreamDestroy(g.aux[i]);
    for (int i = 0; i < MAX_FLIGHT; i++) if (g.streams[i]) (void)hipStreamDestroy(g.streams[i]);
    g = Ctx();
}

extern "C" int mirt_set_profiling(int on)
{
    int rc;
    if ((rc = need_init())) return rc;
    g.profiling = on != 0;
    return MIRT_OK;
}

extern "C" int mirt_sync(void)
{
    int rc;
    if ((rc = need_init())) return rc;
    HIP_TRY(sync_all());
    // A rasteriser frame whose row tables were sized from a cached count reports a table that turned out too small in
    // counters[1] (it would have dropped the rows of the triangles that did not fit): surface it here, where the caller
    // waits for its frames, instead of presenting such a frame silently.
    if (g.raster_since_sync) {
        g.raster_since_sync = false;
        for (RasterScratch &R : g.raster)
            if (R.counters) {
                uint32_t c[2] = { 0, 0 };
                HIP_TRY(hipMemcpy(c, R.counters, sizeof c, hipMemcpyDeviceToHost));
                if (c[1]) {
                    R.sizing_valid = false;
                    return fail(MIRT_ERR_HIP, "rasteriser: the row tables (%zu rows) were too small for a frame; its sizing cache is dropped, render the frame again", R.cap_rows);
                }
            }
    }
    return MIRT_OK;
}

extern "C" void *mirt_stream(void) { return g.init ? (void *)g.stream : nullptr; }

extern "C" int mirt_set_frames_in_flight(int frames)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (frames < 1 || frames > MAX_FLIGHT) return fail(MIRT_ERR_INVALID_ARGUMENT, "frames in flight must be 1 .. %d, not %d", MAX_FLIGHT, frames);
    HIP_TRY(sync_all());
    if (g.d_culled && g.n > 0) {                 // every stream's copy of the cull flags starts from the most recent ones
        for (int h = 0; h < MAX_FLIGHT; h++)
            if (h != g.culled_latest) {
                HIP_TRY(hipMemcpy(g.d_culled + (size_t)h * g.n, g.d_culled + (size_t)g.culled_latest * g.n, (size_t)g.n, hipMemcpyDeviceToDevice));
                g.culled_ver[h] = g.culled_ver[g.culled_latest];
            }
        HIP_TRY(hipDeviceSynchronize());         // (null-stream copies: landed before a frame on one of our streams reads the flags)
    }
    g.in_flight = frames;
    g.si = frames - 1;                           // the first call takes streams[0]
    g.stream = g.streams[0];
    return MIRT_OK;
}

// ---- host surfaces ------------------------------------------------------------------------------------

extern "C" int mirt_surface_register(void *pixels, size_t bytes)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!pixels || bytes == 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "surface must not be NULL / empty");
    for (Ctx::HostSurface &r : g.surf)
        if (r.host == pixels && r.bytes == bytes) return MIRT_OK;
    Ctx::HostSurface *slot = nullptr;
    for (Ctx::HostSurface &r : g.surf) if (!r.host) { slot = &r; break; }
    if (!slot) return fail(MIRT_ERR_INVALID_ARGUMENT, "at most %d surfaces can be registered at a time", (int)(sizeof g.surf / sizeof g.surf[0]));
    hipError_t e = hipHostRegister(pixels, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(MIRT_ERR_HIP, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    void *dev = nullptr;
    e = hipHostGetDevicePointer(&dev, pixels, 0);
    if (e != hipSuccess || !dev) { (void)hipHostUnregister(pixels); (void)hipGetLastError(); return fail(MIRT_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    slot->host = static_cast<char *>(pixels); slot->dev = static_cast<char *>(dev); slot->bytes = bytes;
    return MIRT_OK;
}

extern "C" int mirt_surface_unregister(void *pixels)
{
    int rc;
    if ((rc = need_init())) return rc;
    for (Ctx::HostSurface &r : g.surf)
        if (r.host && r.host == pixels) {
            HIP_TRY(sync_all());
            (void)hipHostUnregister(r.host);
            r = Ctx::HostSurface();
            return MIRT_OK;
        }
    return fail(MIRT_ERR_INVALID_ARGUMENT, "surface %p is not registered", pixels);
}

// ---- scene ------------------------------------------------------------------------------------------

extern "C" int mirt_scene_upload(const float *tris15, const uint8_t *culled, int n)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!tris15 || n < 1) return fail(MIRT_ERR_INVALID_ARGUMENT, "scene needs at least one triangle (n = %d)", n);
    HIP_TRY(sync_all());
    g.n = 0;
    if ((rc = dev_realloc(&g.d_tris, (size_t)n * 15))) return rc;
    if ((rc = dev_realloc(&g.d_culled, (size_t)MAX_FLIGHT * n))) return rc;
    for (RtScratch &S : g.rt) { S.bin_key_valid = false; S.have_known = false; S.count_pending = false; }   // tables and pair counts belong to the old scene
    for (RtScratch &S : g.rt_lt) { S.bin_key_valid = false; S.have_known = false; S.count_pending = false; }
    for (RtScratch &S : g.rt) for (uint64_t &k : S.hist_key) k = 0;                                          // ... and so do the cost histograms
    HIP_TRY(hipMemcpy(g.d_tris, tris15, (size_t)n * 15 * sizeof(float), hipMemcpyHostToDevice));
    g.cull_calls++;
    for (int h = 0; h < MAX_FLIGHT; h++) {
        if (culled) HIP_TRY(hipMemcpy(g.d_culled + (size_t)h * n, culled, (size_t)n, hipMemcpyHostToDevice));
        else HIP_TRY(hipMemset(g.d_culled + (size_t)h * n, 0, (size_t)n));
        g.culled_ver[h] = g.cull_calls;
    }
    if ((rc = dev_realloc(&g.d_geo, (size_t)n))) return rc;
    if ((rc = dev_realloc(&g.d_shade, (size_t)n))) return rc;
    // the copies and fills above ran on the null stream, which the library's non-blocking streams are not ordered with (and a
    // copy from pageable memory may return once the source has been staged): everything has landed before a kernel reads it
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_geo_table, dim3((n + 255) / 256), dim3(256), 0, g.stream, g.d_tris, n, g.d_geo, g.d_shade);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g.stream));
    g.lc.valid = false;
    g.scene_finite = true;
    for (size_t i = 0; i < (size_t)n * 15 && g.scene_finite; i++)
        if (!(fabsf(tris15[i]) < 1.0e8f)) g.scene_finite = false;   // generous: |coord| < 1e8 keeps cross products < 1e18
    for (int c = 0; c < 3; c++) { g.bbox_lo[c] = INFINITY; g.bbox_hi[c] = -INFINITY; }
    for (size_t t = 0; t < (size_t)n; t++)
        for (int v = 0; v < 9; v++) {
            const float x = tris15[t * 15 + v];
            g.bbox_lo[v % 3] = fminf(g.bbox_lo[v % 3], x); g.bbox_hi[v % 3] = fmaxf(g.bbox_hi[v % 3], x);
        }
    g.n = n;
    g.scene_version++;
    return MIRT_OK;
}

extern "C" int mirt_scene_set_culled(const uint8_t *culled, int n)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (g.n <= 0) return fail(MIRT_ERR_NO_SCENE, "no scene uploaded (mirt_scene_upload)");
    if (n != g.n) return fail(MIRT_ERR_INVALID_ARGUMENT, "cull array has %d entries, scene has %d triangles", n, g.n);
    HIP_TRY(sync_all());
    g.cull_calls++;
    for (int h = 0; h < MAX_FLIGHT; h++) {
        if (culled) HIP_TRY(hipMemcpyAsync(g.d_culled + (size_t)h * n, culled, (size_t)n, hipMemcpyHostToDevice, g.stream));
        else HIP_TRY(hipMemsetAsync(g.d_culled + (size_t)h * n, 0, (size_t)n, g.stream));
        g.culled_ver[h] = g.cull_calls;
    }
    HIP_TRY(hipStreamSynchronize(g.stream));
    g.cull_version++;
    return MIRT_OK;
}

// Orders a WRITE into cull-flag copy `dst`, about to be queued on stream `st`, behind every copy OUT of it that another stream
// still has pending (a rasterised frame that brought the latest flags over to its own copy): a bit per copy and stream says
// which streams have read `dst` since the last writer waited; a stream's event is re-recorded behind each of its reads, and a
// stream runs in order, so waiting for its latest record covers the earlier ones.  Both writers come here: the cull kernel
// (mirt_cull_device) and the hand-over copy of raster_enqueue -- with three or four frames in flight the latter can overwrite a
// copy that a lagging stream is still reading (advisor finding of round 3).
static int cull_copy_wait_readers(int dst, hipStream_t st)
{
    for (int r = 0; r < MAX_FLIGHT; r++)
        if ((g.cull_read_src[r] >> dst & 1u) && r != dst) {
            HIP_TRY(hipStreamWaitEvent(st, g.ev_cull_read[r], 0));
            g.cull_read_src[r] &= ~(1u << dst);
        }
    return MIRT_OK;
}

// The cull step on the device, for the uploaded scene: no host copy of the flags in either direction.
extern "C" int mirt_cull_device(const mirt_view *view, int flags)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!view) return fail(MIRT_ERR_INVALID_ARGUMENT, "view must not be NULL");
    if (g.n <= 0) return fail(MIRT_ERR_NO_SCENE, "no scene uploaded (mirt_scene_upload)");
    if (view->width <= 0 || view->height <= 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "frame %d x %d", view->width, view->height);
    CullParams cp;
    cull_setup(view, flags, &cp);
    // The flags belong to the NEXT rasterised frame (the reference culls in Update(), right before Draw()): with several frames
    // in flight they go into the copy of d_culled that belongs to the stream the next call will take, on that stream, in order
    // in front of it -- the frames still running on the other streams keep their own flags, nothing waits for anything.  Should
    // the next rasterised frame land on another stream after all (a ray-traced frame came in between), raster_enqueue brings
    // the flags over (culled_ver tells).
    const int half = next_si();
    hipStream_t st = g.streams[half];
    (void)hipGetLastError();                     // drop a stale error of another HIP user in this thread (see call_begin)
    if ((rc = cull_copy_wait_readers(half, st))) return rc;   // a frame of another stream may still be copying this copy's previous flags
    hipLaunchKernelGGL(k_cull, dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, st, g.d_tris, g.n, cp, g.d_culled + (size_t)half * g.n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g.ev_order[half], st));
    g.culled_latest = half;
    g.culled_ver[half] = ++g.cull_calls;
    g.cull_version++;
    return MIRT_OK;
}

extern "C" int mirt_scene_get_culled(uint8_t *culled, int n)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (g.n <= 0) return fail(MIRT_ERR_NO_SCENE, "no scene uploaded (mirt_scene_upload)");
    if (!culled || n != g.n) return fail(MIRT_ERR_INVALID_ARGUMENT, "cull array has %d entries, scene has %d triangles", n, g.n);
    HIP_TRY(sync_all());
    HIP_TRY(hipMemcpy(culled, g.d_culled + (size_t)g.culled_latest * g.n, (size_t)n, hipMemcpyDeviceToHost));
    return MIRT_OK;
}

extern "C" int mirt_scene_size(void) { return g.init ? g.n : 0; }

extern "C" int mirt_set_depth_of_field(int kernel_size, float focal_length)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (kernel_size > 64) return fail(MIRT_ERR_INVALID_ARGUMENT, "DOF kernel size %d exceeds 64 (the reference uses 8)", kernel_size);
    g.dof_k = kernel_size > 1 ? kernel_size : 0;
    g.dof_focal = focal_length;
    return MIRT_OK;
}

extern "C" int mirt_set_antialiasing(int samples)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (samples > 8) return fail(MIRT_ERR_INVALID_ARGUMENT, "AA samples %d exceed 8 (the reference uses 3)", samples);
    g.aa = samples > 1 ? samples : 1;
    return MIRT_OK;
}

extern "C" int mirt_set_soft_shadows(int samples, const float *positions, int npositions)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (samples <= 1) { g.soft_samples = 1; g.soft_npos = 0; return MIRT_OK; }
    if (samples > MIRT_MAX_LIGHTS) return fail(MIRT_ERR_INVALID_ARGUMENT, "samples %d exceeds %d", samples, MIRT_MAX_LIGHTS);
    if (!positions || npositions < samples || npositions > MIRT_MAX_LIGHTS)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "need between %d and %d jittered positions, got %d", samples, MIRT_MAX_LIGHTS, npositions);
    memcpy(g.soft_pos, positions, sizeof(float) * 3 * (size_t)npositions);
    g.soft_samples = samples;
    g.soft_npos = npositions;
    return MIRT_OK;
}

// ---- ray tracer -------------------------------------------------------------------------------------

extern "C" int mirt_raytrace_device_ex(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                       int mode, int y0, int y1, int row_origin, void *d_xrgb, int pitch_bytes,
                                       void *d_rgb, void *d_index, void *d_distance, void *d_position)
{
    if (g.init && g.dof_k > 1)
        return render_with_dof(view, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb, d_index, nullptr, false,
                               [&](int ry0, int ry1, void *x, void *rgb, void *fd, void *idx, void *) {
                                   return rt_enqueue(view, lights, nlights, indirect, mode, ry0, ry1, ry0, x, view->width * 4, rgb, idx, fd,
                                                     d_distance, d_position);
                               });
    return rt_enqueue(view, lights, nlights, indirect, mode, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb, d_index, nullptr,
                      d_distance, d_position);
}

extern "C" int mirt_raytrace_device(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                    int mode, int y0, int y1, int row_origin, void *d_xrgb, int pitch_bytes,
                                    void *d_rgb, void *d_index)
{
    return mirt_raytrace_device_ex(view, lights, nlights, indirect, mode, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb, d_index,
                                   nullptr, nullptr);
}

extern "C" int mirt_raytrace(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                             int mode, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb, int32_t *out_index)
{
    return mirt_raytrace_ex(view, lights, nlights, indirect, mode, out_xrgb, pitch_bytes, out_rgb, out_index, nullptr, nullptr);
}

extern "C" int mirt_raytrace_ex(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                int mode, uint32_t *out_xrgb, int pitch_bytes, float *out_rgb, int32_t *out_index,
                                float *out_distance, float *out_position)
{
    int rc;
    if ((rc = need_init())) return rc;
    if ((rc = check_view(view, lights, nlights, indirect))) return rc;
    if (!out_xrgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "out_xrgb must not be NULL");
    if (pitch_bytes < view->width * 4 || (pitch_bytes & 3)) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, view->width);
    const int W = view->width, H = view->height;
    const size_t px = (size_t)W * H;
    // closestIntersections[].distance shares the depth staging plane of the rasteriser entry point; .position gets its own
    if ((rc = ensure_staging(px, out_rgb != nullptr, out_index != nullptr, out_distance != nullptr, out_position != nullptr))) return rc;
    char *alias = registered_alias(out_xrgb, (size_t)pitch_bytes, H);
    const bool direct = alias && host_direct();
    if ((rc = mirt_raytrace_device_ex(view, lights, nlights, indirect, mode, 0, H, 0, direct ? (void *)alias : g.d_xrgb, direct ? pitch_bytes : W * 4,
                                      out_rgb ? g.d_rgb : nullptr, out_index ? g.d_index : nullptr,
                                      out_distance ? g.d_zinv : nullptr, out_position ? g.d_pos : nullptr))) return rc;
    if (!direct && (rc = copy_plane_interior(out_xrgb, pitch_bytes, g.d_xrgb, W * 4, W, H))) return rc;
    if (out_rgb) HIP_TRY(hipMemcpyAsync(out_rgb, g.d_rgb, px * 12, hipMemcpyDeviceToHost, g.stream));
    if (out_index) HIP_TRY(hipMemcpyAsync(out_index, g.d_index, px * 4, hipMemcpyDeviceToHost, g.stream));
    if (out_distance) HIP_TRY(hipMemcpyAsync(out_distance, g.d_zinv, px * 4, hipMemcpyDeviceToHost, g.stream));
    if (out_position) HIP_TRY(hipMemcpyAsync(out_position, g.d_pos, px * 12, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return MIRT_OK;
}

// ---- asynchronous delivery into a registered host surface --------------------------------------------------------------
// Render into one of two library-owned planes, then ONE stream-ordered DMA copy into the pinned surface; no host sync.  With
// two frames in flight the copy of frame i (the DMA engine) runs while frame i + 1 renders, so a loop that presents one
// surface while the next one is drawn moves frames at the rate of the link alone.
static int async_plane(size_t px, void **plane)
{
    if (px > g.async_cap_px) {
        HIP_TRY(sync_all());                                 // frames in flight may still read the planes
        for (void *&p : g.d_async) { if (p) (void)hipFree(p); p = nullptr; }
        g.async_cap_px = 0;
        for (void *&p : g.d_async)
            if (hipMalloc(&p, px * 4) != hipSuccess) { p = nullptr; return fail(MIRT_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes) for an asynchronous frame", px * 4); }
        g.async_cap_px = px;
    }
    *plane = g.d_async[next_si()];                           // the plane of the stream this frame is about to take
    return MIRT_OK;
}

static int async_target(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect, uint32_t *out_xrgb, int pitch_bytes, char **alias)
{
    int rc;
    if ((rc = need_init())) return rc;
    if ((rc = check_view(view, lights, nlights, indirect))) return rc;
    if (!out_xrgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "out_xrgb must not be NULL");
    if (pitch_bytes < view->width * 4 || (pitch_bytes & 3)) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, view->width);
    *alias = registered_alias(out_xrgb, (size_t)pitch_bytes, view->height);
    if (!*alias) return fail(MIRT_ERR_INVALID_ARGUMENT, "an asynchronous frame needs a surface registered with mirt_surface_register (pageable memory cannot take a stream-ordered copy)");
    return MIRT_OK;
}

extern "C" int mirt_raytrace_async(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                   int mode, uint32_t *out_xrgb, int pitch_bytes)
{
    int rc;
    char *alias = nullptr;
    if ((rc = async_target(view, lights, nlights, indirect, out_xrgb, pitch_bytes, &alias))) return rc;
    const int W = view->width, H = view->height;
    if (host_direct())
        return mirt_raytrace_device_ex(view, lights, nlights, indirect, mode, 0, H, 0, alias, pitch_bytes, nullptr, nullptr, nullptr, nullptr);
    void *plane = nullptr;
    if ((rc = async_plane((size_t)W * H, &plane))) return rc;
    if ((rc = mirt_raytrace_device_ex(view, lights, nlights, indirect, mode, 0, H, 0, plane, W * 4, nullptr, nullptr, nullptr, nullptr))) return rc;
    return copy_plane_interior(out_xrgb, pitch_bytes, plane, W * 4, W, H);      // on the stream the frame was queued on
}

// ---- rasteriser -------------------------------------------------------------------------------------

static int raster_enqueue(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                          int y0, int y1, int row_origin, void *d_xrgb, int pitch_bytes, void *d_rgb, void *d_zinv,
                          void *d_index, void *d_fd = nullptr)
{
    int rc;
    if ((rc = need_init())) return rc;
    if ((rc = check_view(view, lights, nlights, indirect))) return rc;
    if (g.n <= 0) return fail(MIRT_ERR_NO_SCENE, "no scene uploaded (mirt_scene_upload)");
    if (!d_xrgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "xrgb output must not be NULL");
    if (y0 < 0 || y1 > view->height || y0 > y1) return fail(MIRT_ERR_INVALID_ARGUMENT, "row band [%d,%d) outside [0,%d)", y0, y1, view->height);
    if (pitch_bytes < view->width * 4 || (pitch_bytes & 3)) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, view->width);

    // A rasteriser frame touches the scene (read only) and its stream's own scratch and depth-of-field planes, so frames
    // may overlap (call_begin).
    call_begin();
    RasterScratch &scratch = g.raster[g.si];
    if (g.culled_ver[g.si] != g.culled_ver[g.culled_latest]) {
        // the most recent cull flags sit in another stream's copy (mirt_cull_device wrote them for the call it expected next,
        // and a ray-traced frame took that turn): bring them over, ordered after the cull kernel
        const int from = g.culled_latest;
        HIP_TRY(hipStreamWaitEvent(g.stream, g.ev_order[from], 0));
        if ((rc = cull_copy_wait_readers(g.si, g.stream))) return rc;   // ... and after any stream still copying OUT of this stream's copy
        HIP_TRY(hipMemcpyAsync(g.d_culled + (size_t)g.si * g.n, g.d_culled + (size_t)from * g.n, (size_t)g.n, hipMemcpyDeviceToDevice, g.stream));
        g.culled_ver[g.si] = g.culled_ver[from];
        HIP_TRY(hipEventRecord(g.ev_cull_read[g.si], g.stream));       // (a later cull step into copy `from` must not overtake this read)
        g.cull_read_src[g.si] |= 1u << from;
    }
    g.pending_is_rt = false;
    if (y1 == y0) { call_end(); return MIRT_OK; }

    RasterFrame f;
    memset(&f, 0, sizeof f);
    f.tris15 = g.d_tris;
    f.culled = g.d_culled + (size_t)g.si * g.n;
    f.n = g.n;
    memcpy(f.cam, view->pos, 12);
    memcpy(f.rot, view->rot, 36);
    mat3_inverse(view->rot, f.invrot);        // glm::inverse(cameraRot), hoisted out of PixelShader (rasteriser.cpp:559)
    f.focal = view->focal;
    f.W = view->width; f.H = view->height;
    f.nlights = nlights;
    for (int k = 0; k < nlights; k++) {
        memcpy(f.lpos[k], lights[k].pos, 12);
        for (int c = 0; c < 3; c++) f.lcol[k][c] = lights[k].color[c] * lights[k].intensity;   // rasteriser.cpp:576
    }
    f.lights_in_range = 1;
    for (int k = 0; k < nlights; k++) f.lights_in_range &= light_colour_in_range(f.lcol[k]) ? 1 : 0;
    memcpy(f.indirect, indirect, 12);
    f.y0 = y0; f.y1 = y1; f.row_origin = row_origin;
    f.xrgb = static_cast<uint32_t *>(d_xrgb);
    f.pitch_words = pitch_bytes / 4;
    f.rgb = static_cast<float *>(d_rgb);
    f.zinv = static_cast<float *>(d_zinv);
    f.index = static_cast<int32_t *>(d_index);
    f.fd = static_cast<float *>(d_fd);
    f.focal_plane = g.dof_focal;
    if ((rc = raster_scratch_ensure(scratch, g.n, view->width, y1 - y0))) return fail(rc, "raster scratch allocation failed");
    g.raster_since_sync = true;
    {
        static const int edge_env = [] { const char *e = getenv("MIRT_EDGE_SEGMENTS"); return e ? atoi(e) : -1; }();
        f.edge_segments = edge_env >= 0 ? edge_env : (g.in_flight <= 2 ? 1 : 0);       // (2: tests -- a spoilt prediction in every chain)
    }
    if ((rc = launch_raster(f, scratch, g.scene_version * 0x9E3779B97F4A7C15ull + g.cull_version, g.stream, g.profiling ? &g.ev[EV_K0] : nullptr,
                            g.profiling ? g.ev_used : nullptr)))
        return fail(rc, "rasteriser launch failed: %s", hipGetErrorString(hipGetLastError()));
    call_end();
    return MIRT_OK;
}

extern "C" int mirt_rasterise_device(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                     int y0, int y1, int row_origin, void *d_xrgb, int pitch_bytes, void *d_rgb,
                                     void *d_zinv, void *d_index)
{
    if (g.init && g.dof_k > 1)
        return render_with_dof(view, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb, d_index, d_zinv, true,
                               [&](int ry0, int ry1, void *x, void *rgb, void *fd, void *idx, void *zinv) {
                                   return raster_enqueue(view, lights, nlights, indirect, ry0, ry1, ry0, x, view->width * 4, rgb, zinv, idx, fd);
                               });
    return raster_enqueue(view, lights, nlights, indirect, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb, d_zinv, d_index);
}

extern "C" int mirt_rasterise(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                              uint32_t *out_xrgb, int pitch_bytes, float *out_rgb, float *out_zinv, int32_t *out_index)
{
    int rc;
    if ((rc = need_init())) return rc;
    if ((rc = check_view(view, lights, nlights, indirect))) return rc;
    if (!out_xrgb) return fail(MIRT_ERR_INVALID_ARGUMENT, "out_xrgb must not be NULL");
    if (pitch_bytes < view->width * 4 || (pitch_bytes & 3)) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, view->width);
    const int W = view->width, H = view->height;
    const size_t px = (size_t)W * H;
    if ((rc = ensure_staging(px, out_rgb != nullptr, out_index != nullptr, out_zinv != nullptr))) return rc;
    char *alias = registered_alias(out_xrgb, (size_t)pitch_bytes, H);
    const bool direct = alias && host_direct();
    if ((rc = mirt_rasterise_device(view, lights, nlights, indirect, 0, H, 0, direct ? (void *)alias : g.d_xrgb, direct ? pitch_bytes : W * 4,
                                    out_rgb ? g.d_rgb : nullptr, out_zinv ? g.d_zinv : nullptr, out_index ? g.d_index : nullptr))) return rc;
    // the rasteriser's Update() paints the whole surface (rasteriser.cpp:190): every word is written
    if (!direct) HIP_TRY(hipMemcpy2DAsync(out_xrgb, pitch_bytes, g.d_xrgb, (size_t)W * 4, (size_t)W * 4, H, hipMemcpyDeviceToHost, g.stream));
    if (out_rgb) HIP_TRY(hipMemcpyAsync(out_rgb, g.d_rgb, px * 12, hipMemcpyDeviceToHost, g.stream));
    if (out_zinv) HIP_TRY(hipMemcpyAsync(out_zinv, g.d_zinv, px * 4, hipMemcpyDeviceToHost, g.stream));
    if (out_index) HIP_TRY(hipMemcpyAsync(out_index, g.d_index, px * 4, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return MIRT_OK;
}

extern "C" int mirt_rasterise_async(const mirt_view *view, const mirt_light *lights, int nlights, const float *indirect,
                                    uint32_t *out_xrgb, int pitch_bytes)
{
    int rc;
    char *alias = nullptr;
    if ((rc = async_target(view, lights, nlights, indirect, out_xrgb, pitch_bytes, &alias))) return rc;
    const int W = view->width, H = view->height;
    if (host_direct())
        return mirt_rasterise_device(view, lights, nlights, indirect, 0, H, 0, alias, pitch_bytes, nullptr, nullptr, nullptr);
    void *plane = nullptr;
    if ((rc = async_plane((size_t)W * H, &plane))) return rc;
    if ((rc = mirt_rasterise_device(view, lights, nlights, indirect, 0, H, 0, plane, W * 4, nullptr, nullptr, nullptr))) return rc;
    // the rasteriser's Update() paints the whole surface (rasteriser.cpp:190): every word is written
    HIP_TRY(hipMemcpy2DAsync(out_xrgb, pitch_bytes, plane, (size_t)W * 4, (size_t)W * 4, H, hipMemcpyDeviceToHost, g.stream));
    return MIRT_OK;
}

// ---- statistics -------------------------------------------------------------------------------------

extern "C" int mirt_get_stats(mirt_stats *out)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!out) return fail(MIRT_ERR_INVALID_ARGUMENT, "out must not be NULL");
    if (g.stats_pending) {
        HIP_TRY(hipStreamSynchronize(g.stats_stream));               // the stream the last call ran on
        float ms = 0.0f;
        g.stats.gpu_ms = 0.0f;
        if (g.call_timed && hipEventElapsedTime(&ms, g.ev[EV_CALL0], g.ev[EV_CALL1]) == hipSuccess) g.stats.gpu_ms = ms;
        for (int k = 0; k < 8; k++) {
            g.stats.kernel_ms[k] = 0.0f;
            if (g.profiling && g.ev_used[k] && hipEventElapsedTime(&ms, g.ev[EV_K0 + 2 * k], g.ev[EV_K0 + 2 * k + 1]) == hipSuccess)
                g.stats.kernel_ms[k] = ms;
        }
        if (g.pending_is_rt && g.pending_empty) {
            g.stats.primary_rays = g.stats.shadow_rays = g.stats.tests = 0;
        } else if (g.pending_is_rt) {
            static unsigned long long shard[HIT_SHARDS * HIT_SHARD_STRIDE];
            HIP_TRY(hipMemcpy(shard, g.d_hits, sizeof shard, hipMemcpyDeviceToHost));
            unsigned long long hits = 0, tests = 0, cands = 0, sp = 0, ss = 0, dr = 0;
            for (int i = 0; i < HIT_SHARDS; i++) {
                const unsigned long long *w = shard + i * HIT_SHARD_STRIDE;
                hits += w[0]; tests += w[1]; cands += w[2]; sp += w[3]; ss += w[4]; dr += w[5];
            }
            g.stats.tests = tests;
            g.stats.candidates = cands;
            g.stats.steps_primary = sp; g.stats.steps_shadow = ss; g.stats.drains = dr;
#ifdef MIRT_TR_TIMING
            {   // (experiments, rt_trace.hip built with -DMIRT_TR_TIMING: the shard words carry the waves' time per segment)
                unsigned long long seg[11] = { 0 };
                for (int i = 0; i < HIT_SHARDS; i++) {
                    for (int k = 0; k < 10; k++) seg[k] += shard[i * HIT_SHARD_STRIDE + 3 + k];
                    seg[10] = std::max(seg[10], shard[i * HIT_SHARD_STRIDE + 13]);
                }
                fprintf(stderr, "[mirt timing] Mticks: record %.1f | stage %.1f steps %.1f drains %.1f merge %.1f | light term %.1f offsets+first row %.1f walk %.1f drain %.1f | rest %.1f | longest wave %llu ticks\n",
                        seg[0] * 1e-6, seg[1] * 1e-6, seg[2] * 1e-6, seg[3] * 1e-6, seg[4] * 1e-6, seg[5] * 1e-6, seg[6] * 1e-6, seg[7] * 1e-6, seg[8] * 1e-6, seg[9] * 1e-6, seg[10]);
            }
#endif
            g.stats.primary_rays = g.pending_primary;
            g.stats.shadow_rays = (uint64_t)hits * (uint64_t)g.pending_nlights;
            if (g.stats.mode_used == MIRT_RT_BINNED && g.stats_sel_count) {
                uint32_t nsel = 0;
                HIP_TRY(hipMemcpy(&nsel, g.stats_sel_count, 4, hipMemcpyDeviceToHost));
                g.stats.selected_triangles = nsel;
            }
            if (g.stats.mode_used == MIRT_RT_BRUTE && !g.pending_counted)      // every ray tests every triangle
                g.stats.tests = (g.stats.primary_rays + g.stats.shadow_rays) * (uint64_t)g.n;
            if (g.stats.candidates == 0) g.stats.candidates = g.stats.tests;        // kernels that test every candidate they are offered
        }
        g.stats_pending = false;
        (void)hipGetLastError();                 // an event pair a frame never recorded (no clear needed, ...) leaves hipErrorInvalidHandle
                                                 // behind: do not hand it to the next HIP user of this thread
    }
    *out = g.stats;
    return MIRT_OK;
}

// Per-kernel GPU times of the call BEFORE the last one.
extern "C" int mirt_get_previous_kernel_ms(float *kernel_ms8, float *gpu_ms)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!kernel_ms8) return fail(MIRT_ERR_INVALID_ARGUMENT, "kernel_ms8 must not be NULL");
    if (g.in_flight < 2) return fail(MIRT_ERR_INVALID_ARGUMENT, "the call before the last one keeps its events only with two or more frames in flight (mirt_set_frames_in_flight)");
    HIP_TRY(sync_all());
    const int set = (g.ev_cur + g.in_flight - 1) % g.in_flight;
    float ms = 0.0f;
    if (gpu_ms) *gpu_ms = (g.call_timed_sets[set] && hipEventElapsedTime(&ms, g.ev_sets[set][EV_CALL0], g.ev_sets[set][EV_CALL1]) == hipSuccess) ? ms : 0.0f;
    for (int k = 0; k < 8; k++) {
        kernel_ms8[k] = 0.0f;
        if (g.profiling && g.ev_used_sets[set][k] && hipEventElapsedTime(&ms, g.ev_sets[set][EV_K0 + 2 * k], g.ev_sets[set][EV_K0 + 2 * k + 1]) == hipSuccess)
            kernel_ms8[k] = ms;
    }
    (void)hipGetLastError();
    return MIRT_OK;
}

// ---- several GPUs of one node: frames shard by row bands, one process per GPU ----------------------

extern "C" int mirt_band_of(int rank, int world, int height, int *y0, int *y1)
{
    if (world < 1 || rank < 0 || rank >= world || height < 0 || !y0 || !y1) return fail(MIRT_ERR_INVALID_ARGUMENT, "band of rank %d / %d, %d rows", rank, world, height);
    band_of(rank, world, height, y0, y1);
    return MIRT_OK;
}

extern "C" int mirt_band_plan(int world, int root, int width, int height, int nviews, uint64_t *root_offset, uint64_t *band_offset,
                              uint64_t *bytes, int32_t *peer, int max_pieces)
{
    if (world < 1 || root < 0 || root >= world || width < 1 || height < 0 || nviews < 1 || max_pieces < 0)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "band plan: world %d root %d frame %dx%d views %d", world, root, width, height, nviews);
    std::vector<BandPiece> plan((size_t)std::max(max_pieces, 1));
    const int n = band_gather_plan(world, root, width, height, nviews, plan.data(), max_pieces);
    for (int i = 0; i < n && i < max_pieces; i++) {
        if (root_offset) root_offset[i] = plan[i].root_offset;
        if (band_offset) band_offset[i] = plan[i].band_offset;
        if (bytes) bytes[i] = plan[i].bytes;
        if (peer) peer[i] = plan[i].peer;
    }
    return n;
}

extern "C" int mirt_set_partition(int strip_rows)
{
    if (strip_rows != MIRT_PARTITION_WEIGHTED && (strip_rows < 0 || (strip_rows % BIN_TILE) != 0))
        return fail(MIRT_ERR_INVALID_ARGUMENT, "strip height %d must be 0 (bands), a multiple of %d rows, or MIRT_PARTITION_WEIGHTED", strip_rows, BIN_TILE);
    if (g.init) HIP_TRY(sync_all());
    g.strip_rows = strip_rows;
    return MIRT_OK;
}

extern "C" int mirt_set_cost_histogram(int on)
{
    int rc;
    if ((rc = need_init())) return rc;
    g.want_hist = on != 0;
    return MIRT_OK;
}

// pairs-equivalents a tile costs whatever its list holds (part_weighted_bounds); MIRT_PART_TILE_WEIGHT overrides
static unsigned part_tile_weight()
{
    static const unsigned w = [] { const char *e = getenv("MIRT_PART_TILE_WEIGHT"); long v = e ? atol(e) : -1; return v >= 0 ? (unsigned)v : 12u; }();
    return w;
}

extern "C" int mirt_weighted_bounds(const uint32_t *hist, int hist_rows, int hist_shift, int width, int height, int world, int32_t *bounds)
{
    if (world < 1 || width < 1 || height < 0 || !bounds || hist_rows < 0 || hist_shift < 0 || hist_shift > 16 || (hist_rows > 0 && !hist))
        return fail(MIRT_ERR_INVALID_ARGUMENT, "weighted bounds: world %d frame %dx%d, %d histogram rows, shift %d", world, width, height, hist_rows, hist_shift);
    std::vector<int> b((size_t)world + 1);
    part_weighted_bounds(hist, hist_rows, hist_shift, width, height, world, part_tile_weight(), b.data());
    for (int r = 0; r <= world; r++) bounds[r] = b[(size_t)r];
    return MIRT_OK;
}

// The newest cost histogram of stream 0's ring that was filed under a sharded call <= max_key (0: any) -- waiting for its
// event if it has not fired yet.  NULL when there is none.
static const uint32_t *hist_lookup(uint64_t max_key, int *rows, int *shift)
{
    RtScratch &S = g.rt[0];
    int best = -1;
    for (int i = 0; i < HIST_RING; i++) {
        // ring order breaks ties between copies of one key (outside sharded calls every copy carries the same one): the one filed last
        const int slot = (S.hist_next + HIST_RING - 1 - i) % HIST_RING;
        if (!S.h_hist || S.hist_key[slot] == 0 || (max_key && S.hist_key[slot] > max_key)) continue;
        if (best < 0 || S.hist_key[slot] > S.hist_key[best]) best = slot;
    }
    if (best < 0) return nullptr;
    if (hipEventSynchronize(S.ev_hist[best]) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *rows = S.hist_rows[best]; *shift = S.hist_shift[best];
    return S.h_hist + (size_t)best * SEL_HIST_MAX;
}

extern "C" int mirt_cost_histogram(uint32_t *hist, int max_rows, int *rows, int *shift)
{
    int rc;
    if ((rc = need_init())) return rc;
    int nr = 0, sh = 0;
    const uint32_t *h = hist_lookup(0, &nr, &sh);
    if (!h) { if (rows) *rows = 0; if (shift) *shift = 0; return 0; }
    if (rows) *rows = nr;
    if (shift) *shift = sh;
    for (int i = 0; i < nr && i < max_rows && hist; i++) hist[i] = h[i];
    return nr;
}

// The bands of the sharded call about to be issued (call number g.shard_calls): equal bands, or -- weighted partition -- bands of
// equal estimated cost from the histogram filed under the call before the previous one (or an earlier one).  Every rank of a group
// issues the same calls with the same views, files a histogram in the first binned pass of each call and looks TWO calls back,
// by which time that pass has long run: same histogram on every rank (integer sums over the same triangles), same integer
// arithmetic, same bands -- no exchange.  A rank that skipped a pass because nothing had changed (rt_enqueue_binned: reuse) holds an
// older copy of the SAME view's histogram, i.e. the same numbers.
static void current_bounds(int world, int W, int H, std::vector<int> &bounds)
{
    bounds.assign((size_t)world + 1, 0);
    int nr = 0, sh = 0;
    const uint32_t *h = (g.strip_rows == MIRT_PARTITION_WEIGHTED && g.shard_calls >= 2) ? hist_lookup(g.shard_calls - 1, &nr, &sh) : nullptr;
    // (a histogram of another frame size cannot be this frame's)
    if (h && nr != ((((H + BIN_TILE - 1) / BIN_TILE) - 1) >> sh) + 1) h = nullptr;
    if (h) part_weighted_bounds(h, nr, sh, W, H, world, part_tile_weight(), bounds.data());
    else for (int r = 0; r < world; r++) { int a, b; band_of(r, world, H, &a, &b); bounds[(size_t)r] = a; bounds[(size_t)r + 1] = b; }
}

extern "C" int mirt_partition_bounds(int world, int width, int height, int32_t *bounds)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (world < 1 || width < 1 || height < 0 || !bounds) return fail(MIRT_ERR_INVALID_ARGUMENT, "partition bounds: world %d frame %dx%d", world, width, height);
    std::vector<int> b;
    current_bounds(world, width, height, b);
    for (int r = 0; r <= world; r++) bounds[r] = b[(size_t)r];
    return MIRT_OK;
}

extern "C" int mirt_bounds_plan(int world, int root, int width, int height, int nviews, const int32_t *bounds, uint64_t *root_offset, uint64_t *band_offset,
                                uint64_t *bytes, int32_t *peer, int max_pieces)
{
    if (world < 1 || root < 0 || root >= world || width < 1 || height < 0 || nviews < 1 || max_pieces < 0 || !bounds)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "bounds plan: world %d root %d frame %dx%d views %d", world, root, width, height, nviews);
    for (int r = 0; r < world; r++)
        if (bounds[r] < 0 || bounds[r + 1] < bounds[r] || bounds[r + 1] > height) return fail(MIRT_ERR_INVALID_ARGUMENT, "bounds plan: boundaries must rise from 0 to %d", height);
    std::vector<int> b(bounds, bounds + world + 1);
    std::vector<BandPiece> plan((size_t)std::max(max_pieces, 1));
    const int n = part_gather_plan(world, root, width, height, nviews, 0, plan.data(), max_pieces, b.data());
    for (int i = 0; i < n && i < max_pieces; i++) {
        if (root_offset) root_offset[i] = plan[i].root_offset;
        if (band_offset) band_offset[i] = plan[i].band_offset;
        if (bytes) bytes[i] = plan[i].bytes;
        if (peer) peer[i] = plan[i].peer;
    }
    return n;
}

extern "C" int mirt_partition_segments(int rank, int world, int height, int strip_rows, int32_t *y0, int32_t *y1, int max_segments)
{
    if (world < 1 || rank < 0 || rank >= world || height < 0 || strip_rows < 0 || max_segments < 0) return fail(MIRT_ERR_INVALID_ARGUMENT, "segments of rank %d / %d, %d rows, strips of %d", rank, world, height, strip_rows);
    const int n = part_segments(rank, world, height, strip_rows);
    for (int k = 0; k < n && k < max_segments; k++) {
        int a, b;
        part_segment(rank, world, height, strip_rows, k, &a, &b);
        if (y0) y0[k] = a;
        if (y1) y1[k] = b;
    }
    return n;
}

extern "C" int mirt_partition_plan(int world, int root, int width, int height, int nviews, int strip_rows, uint64_t *root_offset, uint64_t *band_offset,
                                   uint64_t *bytes, int32_t *peer, int max_pieces)
{
    if (world < 1 || root < 0 || root >= world || width < 1 || height < 0 || nviews < 1 || max_pieces < 0 || strip_rows < 0)
        return fail(MIRT_ERR_INVALID_ARGUMENT, "partition plan: world %d root %d frame %dx%d views %d strips %d", world, root, width, height, nviews, strip_rows);
    std::vector<BandPiece> plan((size_t)std::max(max_pieces, 1));
    const int n = part_gather_plan(world, root, width, height, nviews, strip_rows, plan.data(), max_pieces);
    for (int i = 0; i < n && i < max_pieces; i++) {
        if (root_offset) root_offset[i] = plan[i].root_offset;
        if (band_offset) band_offset[i] = plan[i].band_offset;
        if (bytes) bytes[i] = plan[i].bytes;
        if (peer) peer[i] = plan[i].peer;
    }
    return n;
}

extern "C" int mirt_comm_create_id(void *id128)
{
    if (!id128) return fail(MIRT_ERR_INVALID_ARGUMENT, "id must not be NULL");
    if (!comm_create_id(id128)) return fail(MIRT_ERR_HIP, "%s", comm_error(nullptr));
    return MIRT_OK;
}

extern "C" int mirt_comm_init(const void *id128, int rank, int world)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!id128) return fail(MIRT_ERR_INVALID_ARGUMENT, "id must not be NULL");
    if (g.comm) { HIP_TRY(sync_all()); comm_destroy(g.comm); g.comm = nullptr; }
    g.comm = comm_init(id128, rank, world);
    if (!g.comm) return fail(MIRT_ERR_HIP, "%s", comm_error(nullptr));
    if (!g.comm_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&g.comm_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g.ev_rendered, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreateWithFlags(&g.ev_sent[i], hipEventDisableTiming));
    }
    return MIRT_OK;
}

extern "C" int mirt_comm_selfcheck(size_t bytes)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!g.comm) return fail(MIRT_ERR_INVALID_ARGUMENT, "mirt_comm_selfcheck before mirt_comm_init");
    HIP_TRY(sync_all());
    if (!comm_selfcheck(g.comm, bytes, g.comm_stream)) return fail(MIRT_ERR_HIP, "%s", comm_error(g.comm));
    return MIRT_OK;
}

extern "C" int mirt_comm_shutdown(void)
{
    if (!g.init || !g.comm) return MIRT_OK;
    HIP_TRY(sync_all());
    comm_destroy(g.comm);
    g.comm = nullptr;
    return MIRT_OK;
}

// `render(view, y0, y1, row_origin, d_xrgb, pitch)` enqueues one band of one frame on g.stream.
template <class Render>
static int render_sharded(const mirt_view *views, int nviews, int root, void *d_frames, int pitch_bytes, bool writes_every_word, Render render)
{
    int rc;
    if ((rc = need_init())) return rc;
    if (!views || nviews < 1) return fail(MIRT_ERR_INVALID_ARGUMENT, "need at least one view");
    const int W = views[0].width, H = views[0].height;
    for (int v = 1; v < nviews; v++)
        if (views[v].width != W || views[v].height != H) return fail(MIRT_ERR_INVALID_ARGUMENT, "the views of one call must share a frame size");
    if (W < 1 || H < 1) return fail(MIRT_ERR_INVALID_ARGUMENT, "frame size %dx%d", W, H);
    const int world = g.comm ? comm_world(g.comm) : 1, rank = g.comm ? comm_rank(g.comm) : 0;
    if (root < 0 || root >= world) return fail(MIRT_ERR_INVALID_ARGUMENT, "root %d outside [0,%d)", root, world);
    if (rank == root && !d_frames) return fail(MIRT_ERR_INVALID_ARGUMENT, "the root's frame buffer must not be NULL");
    if (rank == root && (pitch_bytes < W * 4 || (pitch_bytes & 3))) return fail(MIRT_ERR_INVALID_ARGUMENT, "pitch %d bytes too small for width %d or not a multiple of 4", pitch_bytes, W);
    if (world > 1 && rank == root && pitch_bytes != W * 4) return fail(MIRT_ERR_INVALID_ARGUMENT, "a sharded frame needs a dense root buffer (pitch == 4 * width)");
    if (world > 1 && g.in_flight != 1) return fail(MIRT_ERR_INVALID_ARGUMENT, "sharded frames overlap through the band buffers: use mirt_set_frames_in_flight(1)");
    const size_t frame_bytes = (size_t)H * (size_t)pitch_bytes;
    // (a sharded call is what cost histograms are filed under, one per call: current_bounds)
    struct CallScope {
        CallScope() { g.in_sharded = true; g.hist_taken = false; }
        ~CallScope() { g.in_sharded = false; g.shard_calls++; }
    } scope;
    if (world == 1) {
        for (int v = 0; v < nviews; v++)
            if ((rc = render(&views[v], 0, H, 0, static_cast<char *>(d_frames) + (size_t)v * frame_bytes, pitch_bytes))) return rc;
        return MIRT_OK;
    }
    // this rank's rows: one contiguous band -- an equal share of the rows, or of the estimated cost (weighted partition) --, or
    // interleaved strips (mirt_set_partition); a band buffer holds the segments of one view back to back
    std::vector<int> wb;
    const int *bounds = nullptr;
    const int strips = g.strip_rows > 0 ? g.strip_rows : 0;
    if (g.strip_rows == MIRT_PARTITION_WEIGHTED) { current_bounds(world, W, H, wb); bounds = wb.data(); }
    const int segs = part_segments(rank, world, H, strips, bounds);
    const size_t band_row = (size_t)W * 4, my_bytes = (size_t)part_rows(rank, world, H, strips, bounds) * band_row;
    const int slot = g.band_slot;
    g.band_slot ^= 1;
    if (rank == root) {
        // the root's own rows are rendered in place; the other ranks' rows arrive straight at their places
        for (int v = 0; v < nviews; v++)
            for (int k = 0; k < segs; k++) {
                int y0, y1;
                part_segment(rank, world, H, strips, k, &y0, &y1, bounds);
                if (y1 > y0 && (rc = render(&views[v], y0, y1, 0, static_cast<char *>(d_frames) + (size_t)v * frame_bytes, pitch_bytes))) return rc;
            }
    } else {
        const size_t need = my_bytes * (size_t)nviews;
        HIP_TRY(hipStreamWaitEvent(g.stream, g.ev_sent[slot], 0));         // the gather that last read this buffer has finished
        if (need > g.band_bytes[slot]) {
            HIP_TRY(hipStreamSynchronize(g.comm_stream));
            if (g.d_band[slot]) (void)hipFree(g.d_band[slot]);
            g.d_band[slot] = nullptr; g.band_bytes[slot] = 0;
            if (hipMalloc(reinterpret_cast<void **>(&g.d_band[slot]), need) != hipSuccess) return fail(MIRT_ERR_OUT_OF_MEMORY, "band buffer (%zu bytes)", need);
            g.band_bytes[slot] = need;
        }
        // the border words the ray tracer never writes travel as 0, whatever the buffer held before (a rasterised batch, a
        // batch of another frame size)
        if (!writes_every_word && need) HIP_TRY(hipMemsetAsync(g.d_band[slot], 0, need, g.stream));
        for (int v = 0; v < nviews; v++) {
            int before = 0;                                  // rows of this view's earlier segments in the band buffer
            for (int k = 0; k < segs; k++) {
                int y0, y1;
                part_segment(rank, world, H, strips, k, &y0, &y1, bounds);
                // (row y of the segment lands at row before + (y - y0) of this view's part of the buffer)
                if (y1 > y0 && (rc = render(&views[v], y0, y1, y0 - before, g.d_band[slot] + (size_t)v * my_bytes, (int)band_row))) return rc;
                before += y1 - y0;
            }
        }
    }
    // the one exchange step: every band to the root, on the communication stream, overlapping the next call's render
    HIP_TRY(hipEventRecord(g.ev_rendered, g.stream));
    HIP_TRY(hipStreamWaitEvent(g.comm_stream, g.ev_rendered, 0));
    const int maxp = part_gather_plan(world, root, W, H, nviews, strips, nullptr, 0, bounds);
    std::vector<BandPiece> plan((size_t)std::max(maxp, 1));
    const int np = part_gather_plan(world, root, W, H, nviews, strips, plan.data(), (int)plan.size(), bounds);
    std::vector<GatherPiece> pieces;
    for (int i = 0; i < np; i++) {
        if (plan[i].bytes == 0) continue;
        if (rank == root) pieces.push_back({ static_cast<char *>(d_frames) + plan[i].root_offset, plan[i].bytes, plan[i].peer });
        else if (plan[i].peer == rank) pieces.push_back({ g.d_band[slot] + plan[i].band_offset, plan[i].bytes, root });
    }
    if (!pieces.empty() && !comm_gather_bands(g.comm, root, pieces.data(), (int)pieces.size(), g.comm_stream))
        return fail(MIRT_ERR_HIP, "%s", comm_error(g.comm));
    HIP_TRY(hipEventRecord(g.ev_sent[slot], g.comm_stream));
    return MIRT_OK;
}

extern "C" int mirt_raytrace_sharded(const mirt_view *views, int nviews, const mirt_light *lights, int nlights, const float *indirect,
                                     int mode, int root, void *d_frames, int pitch_bytes)
{
    return render_sharded(views, nviews, root, d_frames, pitch_bytes, false, [&](const mirt_view *v, int y0, int y1, int origin, void *dst, int pitch) {
        return mirt_raytrace_device(v, lights, nlights, indirect, mode, y0, y1, origin, dst, pitch, nullptr, nullptr);
    });
}

extern "C" int mirt_rasterise_sharded(const mirt_view *views, int nviews, const mirt_light *lights, int nlights, const float *indirect,
                                      int root, void *d_frames, int pitch_bytes)
{
    return render_sharded(views, nviews, root, d_frames, pitch_bytes, true, [&](const mirt_view *v, int y0, int y1, int origin, void *dst, int pitch) {
        return mirt_rasterise_device(v, lights, nlights, indirect, y0, y1, origin, dst, pitch, nullptr, nullptr, nullptr);
    });
}

