// World-size-2 (and 3, 5) test of the band partition and the gather plan of the sharded entry points, on the CPU: one
// process per rank (fork), the bands travel through pipes exactly as the plan lists them (csrc/comm.cpp: band_of,
// band_gather_plan -- the arithmetic mirt_*_sharded hands to RCCL), and the root checks every word of every frame.
#include "../../cpp-raytracer-rasterizer_amd/csrc/comm.hpp"

#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace mirt;

static uint32_t pattern(int view, int y, int x) { return 0x9E3779B9u * (uint32_t)(view + 1) ^ (uint32_t)(y * 40503 + x * 65599 + 17); }

// strip == 0: contiguous bands; > 0: interleaved strips of that many rows; bounds != NULL: bands with explicit boundaries (the
// weighted partition: part_weighted_bounds)
static bool run(int world, int root, int W, int H, int nviews, int strip = 0, const int *bounds = nullptr)
{
    std::vector<BandPiece> plan((size_t)std::max(1, part_gather_plan(world, root, W, H, nviews, strip, nullptr, 0, bounds)));
    const int np = part_gather_plan(world, root, W, H, nviews, strip, plan.data(), (int)plan.size(), bounds);
    std::vector<int> rd(world, -1), wr(world, -1);
    std::vector<pid_t> pids;
    for (int r = 0; r < world; r++) {
        if (r == root) continue;
        int fd[2];
        if (pipe(fd) != 0) return false;
        const pid_t pid = fork();
        if (pid == 0) {                                   // rank r: render my band of every view, send the pieces the plan gives me
            close(fd[0]);
            // the band buffer: this rank's segments of one view back to back, views one after the other
            const int mine = part_rows(r, world, H, strip, bounds), segs = part_segments(r, world, H, strip, bounds);
            std::vector<uint32_t> band((size_t)nviews * mine * W + 1);
            for (int v = 0; v < nviews; v++) {
                int before = 0;
                for (int k = 0; k < segs; k++) {
                    int y0, y1;
                    part_segment(r, world, H, strip, k, &y0, &y1, bounds);
                    for (int y = y0; y < y1; y++)
                        for (int x = 0; x < W; x++) band[((size_t)v * mine + before + (y - y0)) * W + x] = pattern(v, y, x);
                    before += y1 - y0;
                }
            }
            for (int i = 0; i < np; i++)
                if (plan[i].peer == r) {
                    const char *p = reinterpret_cast<const char *>(band.data()) + plan[i].band_offset;
                    size_t left = plan[i].bytes;
                    while (left) { const ssize_t n = write(fd[1], p, left); if (n <= 0) _exit(2); p += n; left -= (size_t)n; }
                }
            close(fd[1]);
            _exit(0);
        }
        close(fd[1]);
        rd[r] = fd[0];
        pids.push_back(pid);
    }
    // the root: its own rows in place, the other bands where the plan puts them
    std::vector<uint32_t> frames((size_t)nviews * H * W, 0xDEADBEEFu);
    for (int k = 0, segs = part_segments(root, world, H, strip, bounds); k < segs; k++) {
        int y0, y1;
        part_segment(root, world, H, strip, k, &y0, &y1, bounds);
        for (int v = 0; v < nviews; v++)
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < W; x++) frames[((size_t)v * H + y) * W + x] = pattern(v, y, x);
    }
    bool ok = true;
    for (int i = 0; i < np && ok; i++) {
        char *p = reinterpret_cast<char *>(frames.data()) + plan[i].root_offset;
        size_t left = plan[i].bytes;
        while (left) { const ssize_t n = read(rd[plan[i].peer], p, left); if (n <= 0) { ok = false; break; } p += n; left -= (size_t)n; }
    }
    for (int r = 0; r < world; r++) if (rd[r] >= 0) close(rd[r]);
    for (pid_t pid : pids) { int st = 0; waitpid(pid, &st, 0); ok = ok && WIFEXITED(st) && WEXITSTATUS(st) == 0; }
    for (int v = 0; v < nviews && ok; v++)
        for (int y = 0; y < H && ok; y++)
            for (int x = 0; x < W; x++)
                if (frames[((size_t)v * H + y) * W + x] != pattern(v, y, x)) { ok = false; fprintf(stderr, "world %d root %d: frame %d (%d,%d) wrong\n", world, root, v, x, y); break; }
    if (strip > 0 || bounds) return ok;
    // the bands tile [0, H) in rank order
    int next = 0;
    for (int r = 0; r < world; r++) { int a, b; band_of(r, world, H, &a, &b); ok = ok && a == next && b >= a && b - a <= H / world + 1; next = b; }
    return ok && next == H;
}

int main()
{
    const int cases[][5] = { { 2, 0, 64, 48, 1 }, { 2, 1, 33, 7, 3 }, { 3, 0, 20, 10, 2 }, { 3, 2, 17, 2, 1 }, { 5, 3, 9, 23, 4 }, { 1, 0, 8, 8, 2 }, { 4, 0, 5, 3, 2 } };
    for (const auto &c : cases)
        if (!run(c[0], c[1], c[2], c[3], c[4])) { printf("FAILED world %d root %d %dx%d views %d\n", c[0], c[1], c[2], c[3], c[4]); return 1; }
    const int strips[][6] = { { 2, 0, 16, 40, 1, 8 }, { 3, 1, 7, 100, 2, 16 }, { 5, 4, 3, 9, 1, 8 }, { 4, 2, 5, 64, 3, 64 }, { 8, 0, 6, 432, 2, 64 } };
    for (const auto &c : strips)
        if (!run(c[0], c[1], c[2], c[3], c[4], c[5])) { printf("FAILED strips world %d root %d %dx%d views %d strip %d\n", c[0], c[1], c[2], c[3], c[4], c[5]); return 1; }
    // the weighted partition: boundaries from a cost histogram with a peak in the middle of the frame, then the same gather
    for (const auto &c : strips) {
        const int world = c[0], root = c[1], W = c[2], H = c[3], nviews = c[4];
        const int tile_rows = (H + 7) / 8;
        std::vector<uint32_t> hist((size_t)tile_rows);
        for (int j = 0; j < tile_rows; j++) { const int d = j - tile_rows / 2; hist[(size_t)j] = (uint32_t)(100000 / (1 + d * d)); }
        std::vector<int> bounds((size_t)world + 1);
        part_weighted_bounds(hist.data(), tile_rows, 0, W, H, world, 12u, bounds.data());
        bool ok = bounds[0] == 0 && bounds[(size_t)world] == H;
        for (int r = 0; r < world; r++) ok = ok && bounds[(size_t)r] <= bounds[(size_t)r + 1] && (bounds[(size_t)r + 1] % 8 == 0 || bounds[(size_t)r + 1] == H);
        if (!ok || !run(world, root, W, H, nviews, 0, bounds.data())) { printf("FAILED weighted world %d root %d %dx%d views %d\n", world, root, W, H, nviews); return 1; }
    }
    printf("ok\n");
    return 0;
}
