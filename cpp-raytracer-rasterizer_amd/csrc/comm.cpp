// comm.cpp -- the one exchange step of a sharded frame: every rank's row band travels to the root.
//
// Transports:
//   rccl  point-to-point ncclSend / ncclRecv inside one group (each peer -> root on its own xGMI link: a direct gather,
//         not a ring; SURVEY names ncclGather, which RCCL implements as exactly this group).  The library is loaded at run
//         time (dlopen), preferring a copy the process already holds -- a PyTorch process carries its own librccl, and two
//         different copies in one process must not both be live.
//   shm   host-staged loopback through files in /dev/shm: ranks that share ONE GPU (tests on a one-GPU box, CPU-side
//         rehearsal of the control flow).  Synchronous, slow, and never the measured path.
// No HIP kernels here; plain host code.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace mirt {

void band_of(int rank, int world, int height, int *y0, int *y1)
{
    const int base = height / world, rem = height % world;
    *y0 = rank * base + (rank < rem ? rank : rem);
    *y1 = *y0 + base + (rank < rem ? 1 : 0);
}

int part_segments(int rank, int world, int height, int strip_rows, const int *bounds)
{
    if (bounds) return bounds[rank + 1] > bounds[rank] ? 1 : 0;
    if (strip_rows <= 0) { int a, b; band_of(rank, world, height, &a, &b); return b > a ? 1 : 0; }
    const int strips = (height + strip_rows - 1) / strip_rows;
    return strips > rank ? (strips - rank + world - 1) / world : 0;
}

void part_segment(int rank, int world, int height, int strip_rows, int k, int *y0, int *y1, const int *bounds)
{
    if (bounds) { *y0 = bounds[rank]; *y1 = bounds[rank + 1]; return; }
    if (strip_rows <= 0) { band_of(rank, world, height, y0, y1); return; }
    const long long s = (long long)rank + (long long)k * world;       // the k-th strip of this rank
    *y0 = (int)std::min<long long>(s * strip_rows, height);
    *y1 = (int)std::min<long long>((s + 1) * strip_rows, height);
}

int part_rows(int rank, int world, int height, int strip_rows, const int *bounds)
{
    int rows = 0;
    for (int k = 0, n = part_segments(rank, world, height, strip_rows, bounds); k < n; k++) {
        int a, b;
        part_segment(rank, world, height, strip_rows, k, &a, &b, bounds);
        rows += b - a;
    }
    return rows;
}

int part_gather_plan(int world, int root, int width, int height, int nviews, int strip_rows, BandPiece *out, int max_pieces, const int *bounds)
{
    const size_t row = (size_t)width * 4, frame = (size_t)height * row;
    int n = 0;
    for (int r = 0; r < world; r++) {
        if (r == root) continue;
        const int segs = part_segments(r, world, height, strip_rows, bounds);
        const size_t mine = (size_t)part_rows(r, world, height, strip_rows, bounds) * row;      // this rank's rows of ONE view in its band buffer
        for (int v = 0; v < nviews; v++) {
            size_t before = 0;
            for (int k = 0; k < segs; k++) {
                int a, b;
                part_segment(r, world, height, strip_rows, k, &a, &b, bounds);
                const size_t bytes = (size_t)(b - a) * row;
                if (bytes == 0) continue;
                if (n < max_pieces) out[n] = { (size_t)v * frame + (size_t)a * row, (size_t)v * mine + before, bytes, r };
                n++;
                before += bytes;
            }
        }
    }
    return n;
}

// Bands of equal ESTIMATED COST (the repo's counterpart of the reference's `#pragma omp parallel for schedule(auto)` over rows,
// raytracer.cpp:557: rows go where the work is).  hist[c] = estimated (tile, triangle) pairs of coarse tile row c (tile rows
// c << shift .. ((c + 1) << shift) - 1) of the whole frame, as k_prep_select counts them; a tile row's cost = its share of its
// coarse row's pairs + tile_weight per tile (the work a pixel costs whatever its tile's list holds: ray set-up, shading, the
// shadow ray, the store).  Boundaries fall on tile rows (8 pixel rows), rank r gets the tile rows whose running cost lies in
// [r, r + 1) * total / world, every rank at least one tile row while there are enough.  Integer arithmetic only: every rank of
// a group computes the same boundaries from the same histogram.  bounds: world + 1 rows, bounds[0] = 0, bounds[world] = height.
void part_weighted_bounds(const uint32_t *hist, int hist_rows, int shift, int width, int height, int world, unsigned tile_weight, int *bounds)
{
    const int tile = 8, tile_rows = (height + tile - 1) / tile, tiles_x = (width + tile - 1) / tile;
    std::vector<unsigned long long> cost((size_t)std::max(tile_rows, 1), 0ull);
    for (int j = 0; j < tile_rows; j++) {
        const int c = j >> shift;
        const int first = c << shift, last = std::min(tile_rows, (c + 1) << shift);      // the coarse row's tile rows
        const unsigned long long h = (hist && c < hist_rows) ? hist[c] : 0ull;
        cost[(size_t)j] = h / (unsigned long long)std::max(last - first, 1) + (unsigned long long)tile_weight * (unsigned long long)tiles_x + 1ull;
    }
    // prefix[j] = cost of tile rows [0, j)
    std::vector<unsigned long long> prefix((size_t)tile_rows + 1, 0ull);
    for (int j = 0; j < tile_rows; j++) prefix[(size_t)j + 1] = prefix[(size_t)j] + cost[(size_t)j];
    const unsigned long long total = prefix[(size_t)tile_rows];
    bounds[0] = 0;
    int j = 0;
    for (int r = 1; r < world; r++) {
        // the cut (in tile rows) whose running cost is nearest to r / world of the total (compared as prefix * world vs total * r:
        // 128-bit products, no division, no rounding)
        const unsigned __int128 want = (unsigned __int128)total * (unsigned)r;
        while (j < tile_rows && (unsigned __int128)prefix[(size_t)j + 1] * (unsigned)world <= want) j++;
        int cut = j;
        if (j < tile_rows) {
            const unsigned __int128 lo = (unsigned __int128)prefix[(size_t)j] * (unsigned)world, hi = (unsigned __int128)prefix[(size_t)j + 1] * (unsigned)world;
            if (hi - want < want - lo) cut = j + 1;
        }
        // every rank keeps at least one tile row while the frame has enough of them
        const int prev = (bounds[r - 1] + tile - 1) / tile;
        if (tile_rows >= world) cut = std::min(std::max(cut, prev + 1), tile_rows - (world - r));
        cut = std::max(cut, prev);
        bounds[r] = std::min(cut * tile, height);
    }
    bounds[world] = height;
    for (int r = 1; r <= world; r++) bounds[r] = std::max(bounds[r], bounds[r - 1]);
}

int band_gather_plan(int world, int root, int width, int height, int nviews, BandPiece *out, int max_pieces)
{
    return part_gather_plan(world, root, width, height, nviews, 0, out, max_pieces);
}

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;

    bool load()
    {
        if (lib) return true;
        const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (const char *n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;     // a copy the process already holds
        for (const char *n : names) { if (lib) break; lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
        if (!lib) { err = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?"); return false; }
#define MIRT_SYM(field, name)                                                                                  \
        field = reinterpret_cast<decltype(field)>(dlsym(lib, name));                                           \
        if (!field) { err = std::string("librccl.so lacks ") + name; lib = nullptr; return false; }
        MIRT_SYM(GetUniqueId, "ncclGetUniqueId") MIRT_SYM(CommInitRank, "ncclCommInitRank") MIRT_SYM(CommDestroy, "ncclCommDestroy")
        MIRT_SYM(GroupStart, "ncclGroupStart") MIRT_SYM(GroupEnd, "ncclGroupEnd") MIRT_SYM(Send, "ncclSend") MIRT_SYM(Recv, "ncclRecv")
        MIRT_SYM(GetErrorString, "ncclGetErrorString")
#undef MIRT_SYM
        return true;
    }
};

Rccl g_rccl;

bool use_shm()
{
    const char *e = getenv("MIRT_COMM");
    return e && !strcmp(e, "shm");
}

}  // namespace

struct Comm {
    int rank = 0, world = 1;
    bool shm = false;
    ncclComm_t nccl = nullptr;
    std::string name;                       // shm: the group's file prefix
    std::vector<unsigned> seq_out, seq_in;  // shm: messages so far to / from each peer
    std::vector<char> host;                 // shm: staging
    std::string err;
};

const char *comm_error(const Comm *c) { return c ? c->err.c_str() : g_rccl.err.c_str(); }
int comm_rank(const Comm *c) { return c->rank; }
int comm_world(const Comm *c) { return c->world; }

bool comm_create_id(void *id128)
{
    memset(id128, 0, COMM_ID_BYTES);
    if (use_shm()) {
        snprintf(static_cast<char *>(id128), COMM_ID_BYTES, "mirt_%ld_%lld", (long)getpid(),
                 (long long)std::chrono::steady_clock::now().time_since_epoch().count());
        return true;
    }
    if (!g_rccl.load()) return false;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_rccl.err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return false; }
    static_assert(sizeof id.internal == COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, id.internal, COMM_ID_BYTES);
    return true;
}

Comm *comm_init(const void *id128, int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) { g_rccl.err = "rank / world out of range"; return nullptr; }
    Comm *c = new Comm;
    c->rank = rank; c->world = world; c->shm = use_shm();
    if (c->shm) {
        char name[COMM_ID_BYTES + 1];
        memcpy(name, id128, COMM_ID_BYTES); name[COMM_ID_BYTES] = 0;
        c->name = std::string("/dev/shm/") + name;
        c->seq_out.assign(world, 0); c->seq_in.assign(world, 0);
        return c;
    }
    if (!g_rccl.load()) { delete c; return nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id128, COMM_ID_BYTES);
    const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { g_rccl.err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); delete c; return nullptr; }
    return c;
}

void comm_destroy(Comm *c)
{
    if (!c) return;
    if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    delete c;
}

// ---- shm transport: one file per message, published by rename ----------------------------------------------------------
static bool shm_send(Comm *c, const void *dev, size_t bytes, int peer, hipStream_t stream)
{
    if (hipStreamSynchronize(stream) != hipSuccess) { c->err = "shm send: stream sync failed"; return false; }
    c->host.resize(bytes);
    if (hipMemcpy(c->host.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { c->err = "shm send: copy to host failed"; return false; }
    char path[512], tmp[512];
    snprintf(path, sizeof path, "%s_%d_%d_%u", c->name.c_str(), c->rank, peer, c->seq_out[peer]);
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    c->seq_out[peer]++;
    FILE *f = fopen(tmp, "wb");
    if (!f || fwrite(c->host.data(), 1, bytes, f) != bytes) { if (f) fclose(f); c->err = std::string("shm send: cannot write ") + tmp; return false; }
    fclose(f);
    if (rename(tmp, path) != 0) { c->err = std::string("shm send: cannot publish ") + path; return false; }
    return true;
}

static bool shm_recv(Comm *c, void *dev, size_t bytes, int peer, hipStream_t stream)
{
    char path[512];
    snprintf(path, sizeof path, "%s_%d_%d_%u", c->name.c_str(), peer, c->rank, c->seq_in[peer]);
    c->seq_in[peer]++;
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(path, &st) != 0 || (size_t)st.st_size != bytes) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { c->err = std::string("shm recv: timed out waiting for ") + path; return false; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    c->host.resize(bytes);
    FILE *f = fopen(path, "rb");
    if (!f || fread(c->host.data(), 1, bytes, f) != bytes) { if (f) fclose(f); c->err = std::string("shm recv: cannot read ") + path; return false; }
    fclose(f);
    unlink(path);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(dev, c->host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
        c->err = "shm recv: copy to device failed";
        return false;
    }
    return true;
}

bool comm_gather_bands(Comm *c, int root, const GatherPiece *pieces, int npieces, hipStream_t stream)
{
    if (c->shm) {
        for (int i = 0; i < npieces; i++) {
            const GatherPiece &p = pieces[i];
            if (c->rank == root ? !shm_recv(c, p.ptr, p.bytes, p.peer, stream) : !shm_send(c, p.ptr, p.bytes, root, stream)) return false;
        }
        return true;
    }
    ncclResult_t r = g_rccl.GroupStart();
    for (int i = 0; i < npieces && r == ncclSuccess; i++) {
        const GatherPiece &p = pieces[i];
        r = (c->rank == root) ? g_rccl.Recv(p.ptr, p.bytes, ncclUint8, p.peer, c->nccl, stream)
                              : g_rccl.Send(p.ptr, p.bytes, ncclUint8, root, c->nccl, stream);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) { c->err = std::string("RCCL band gather: ") + g_rccl.GetErrorString(r); return false; }
    return true;
}

bool comm_selfcheck(Comm *c, size_t bytes, hipStream_t stream)
{
    if (!bytes) return true;
    std::vector<unsigned char> pattern(bytes), back(bytes);
    for (size_t i = 0; i < bytes; i++) pattern[i] = (unsigned char)((i * 2654435761u) >> 13);
    void *src = nullptr, *dst = nullptr;
    bool ok = hipMalloc(&src, bytes) == hipSuccess && hipMalloc(&dst, bytes) == hipSuccess &&
              hipMemcpy(src, pattern.data(), bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemset(dst, 0, bytes) == hipSuccess &&
              hipDeviceSynchronize() == hipSuccess;      // (null-stream copy and fill: landed before `stream`, a non-blocking one, touches the buffers)
    if (!ok) c->err = "link check: device buffers";
    if (ok && c->shm) ok = shm_send(c, src, bytes, c->rank, stream) && shm_recv(c, dst, bytes, c->rank, stream);
    else if (ok) {
        ncclResult_t r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.Send(src, bytes, ncclUint8, c->rank, c->nccl, stream);
        if (r == ncclSuccess) r = g_rccl.Recv(dst, bytes, ncclUint8, c->rank, c->nccl, stream);
        const ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { c->err = std::string("link check: ") + g_rccl.GetErrorString(r); ok = false; }
    }
    if (ok && (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(back.data(), dst, bytes, hipMemcpyDeviceToHost) != hipSuccess)) {
        c->err = "link check: reading the received bytes back failed";
        ok = false;
    }
    if (ok && memcmp(back.data(), pattern.data(), bytes) != 0) { c->err = "link check: received bytes differ from the bytes sent"; ok = false; }
    (void)hipFree(src); (void)hipFree(dst);
    return ok;
}

}  // namespace mirt
