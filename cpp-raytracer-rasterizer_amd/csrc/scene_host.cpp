// scene_host.cpp -- host-side scene utilities of the C-ABI: the Cornell-box model, the synthetic soup
// generator and the rasteriser's cull step.  Plain C++ (no device code), built with -ffp-contract=off so
// the float arithmetic is the reference's, operation for operation (mirt_math.hpp).
#include "mirt_math.hpp"
#include "../../include/mirt.h"

#include "cull.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mirt;

namespace {

// Triangle::ComputeNormal (raytracer/Source/TestModel.h:26-31): normalize(cross(e2, e1))
void compute_normal(float *t)
{
    const v3 v0 = ld3(t), v1 = ld3(t + 3), v2 = ld3(t + 6);
    st3(t + 9, normalize3(cross3(sub3(v2, v0), sub3(v1, v0))));
}

struct ModelBuilder {
    float *t;
    int n;
    void tri(v3 a, v3 b, v3 c, v3 col)
    {
        float *p = t + 15 * n++;
        st3(p, a); st3(p + 3, b); st3(p + 6, c); st3(p + 12, col);
        compute_normal(p);
    }
    // the five quads of a block, in the order LoadTestModel pushes them (TestModel.h:113-130,146-163)
    void block(v3 A, v3 B, v3 C, v3 D, v3 E, v3 F, v3 G, v3 H, v3 col)
    {
        tri(E, B, A, col); tri(E, F, B, col);
        tri(F, D, B, col); tri(F, H, D, col);
        tri(H, C, D, col); tri(H, G, C, col);
        tri(G, E, C, col); tri(E, A, C, col);
        tri(G, F, E, col); tri(G, H, F, col);
    }
};

// std::mt19937 restated (public algorithm); the soup generator is ours, not the reference's.
struct Mt19937 {
    uint32_t s[624];
    int i;
    explicit Mt19937(uint32_t seed)
    {
        s[0] = seed;
        for (int k = 1; k < 624; k++) s[k] = 1812433253u * (s[k - 1] ^ (s[k - 1] >> 30)) + (uint32_t)k;
        i = 624;
    }
    uint32_t next()
    {
        if (i >= 624) {
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
                s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            i = 0;
        }
        uint32_t y = s[i++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    float unit() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

}  // namespace

// LoadTestModel, raytracer/Source/TestModel.h:51-192
extern "C" int mirt_scene_cornell(float *tris15)
{
    if (!tris15) return MIRT_ERR_INVALID_ARGUMENT;
    const v3 red = V3(0.75f, 0.15f, 0.15f), yellow = V3(0.75f, 0.75f, 0.15f), green = V3(0.15f, 0.75f, 0.15f),
             cyan = V3(0.15f, 0.75f, 0.75f), blue = V3(0.15f, 0.15f, 0.75f), purple = V3(0.75f, 0.15f, 0.75f),
             white = V3(0.75f, 0.75f, 0.75f);
    const float L = 555;
    ModelBuilder m = { tris15, 0 };
    {
        const v3 A = V3(L, 0, 0), B = V3(0, 0, 0), C = V3(L, 0, L), D = V3(0, 0, L);
        const v3 E = V3(L, L, 0), F = V3(0, L, 0), G = V3(L, L, L), H = V3(0, L, L);
        m.tri(C, B, A, green);  m.tri(C, D, B, green);
        m.tri(A, E, C, purple); m.tri(C, E, G, purple);
        m.tri(F, B, D, yellow); m.tri(H, F, D, yellow);
        m.tri(E, F, G, cyan);   m.tri(F, H, G, cyan);
        m.tri(G, D, C, white);  m.tri(G, H, D, white);
    }
    m.block(V3(290, 0, 114), V3(130, 0, 65), V3(240, 0, 272), V3(82, 0, 225),
            V3(290, 165, 114), V3(130, 165, 65), V3(240, 165, 272), V3(82, 165, 225), red);
    m.block(V3(423, 0, 247), V3(265, 0, 296), V3(472, 0, 406), V3(314, 0, 456),
            V3(423, 330, 247), V3(265, 330, 296), V3(472, 330, 406), V3(314, 330, 456), blue);
    const float k = 2 / L;                       // TestModel.h:174 (`2/L` with float L)
    for (int i = 0; i < m.n; i++) {
        float *p = tris15 + 15 * i;
        for (int v = 0; v < 3; v++) {
            float *q = p + 3 * v;
            for (int c = 0; c < 3; c++) { q[c] *= k; q[c] -= 1.0f; }
            q[0] *= -1; q[1] *= -1;
        }
        compute_normal(p);
    }
    return m.n;
}

// SURVEY section 8(d) config 3/5 generator: 12 draws per triangle (centre, edge a, edge b, colour).
extern "C" int mirt_scene_soup(uint32_t seed, int n, float s, float *tris15)
{
    if (!tris15 || n < 0) return MIRT_ERR_INVALID_ARGUMENT;
    Mt19937 g(seed);
    ModelBuilder m = { tris15, 0 };
    for (int i = 0; i < n; i++) {
        float u[12];
        for (int k = 0; k < 12; k++) u[k] = g.unit();
        const v3 c = V3(2.0f * u[0] - 1.0f, 2.0f * u[1] - 1.0f, 2.0f * u[2] - 1.0f);
        const v3 a = V3(s * (2.0f * u[3] - 1.0f), s * (2.0f * u[4] - 1.0f), s * (2.0f * u[5] - 1.0f));
        const v3 b = V3(s * (2.0f * u[6] - 1.0f), s * (2.0f * u[7] - 1.0f), s * (2.0f * u[8] - 1.0f));
        m.tri(c, add3(c, a), add3(c, b), V3(0.15f + 0.6f * u[9], 0.15f + 0.6f * u[10], 0.15f + 0.6f * u[11]));
    }
    return n;
}

// LoadSTL::LoadSTLFile + split (rasteriser/Source/LoadSTL.cpp:17-97): an ASCII STL as the reference reads it -- every
// line containing "outer" is followed by three vertex lines, split at single spaces with empty tokens and the word
// "vertex" dropped, the first three tokens through atof; afterwards every coordinate is multiplied by -scale (the
// reference: 0.05f) and the normal recomputed (Triangle::ComputeNormal).  "facet normal", "endloop", "solid" lines are
// never looked at, exactly as there.  A vertex line with fewer than three tokens is undefined behaviour in the reference
// (out-of-range vector index); here it ends the load with MIRT_ERR_INVALID_ARGUMENT.
extern "C" int mirt_scene_load_stl(const char *path, float scale, const float *colour3, float *tris15, int max_tris)
{
    if (!path || !colour3 || (tris15 && max_tris < 0)) return MIRT_ERR_INVALID_ARGUMENT;
    FILE *f = std::fopen(path, "rb");
    if (!f) return MIRT_ERR_INVALID_ARGUMENT;
    std::string line;
    auto getline = [&](std::string &out) -> bool {             // std::getline(stream, line): up to '\n', which is dropped
        out.clear();
        int c;
        bool any = false;
        while ((c = std::fgetc(f)) != EOF) {
            any = true;
            if (c == '\n') return true;
            out.push_back((char)c);
        }
        return any;
    };
    auto split = [](const std::string &str, std::vector<std::string> &tok) {      // LoadSTL::split, :84-97
        tok.clear();
        size_t a = 0;
        while (a <= str.size()) {
            size_t b = str.find(' ', a);
            if (b == std::string::npos) b = str.size();
            const std::string t = str.substr(a, b - a);
            if (t.size() > 0 && t != "vertex") tok.push_back(t);
            a = b + 1;
        }
    };
    int n = 0, rc = 0;
    std::vector<std::string> tok;
    while (getline(line)) {
        if (line.find("outer") == std::string::npos) continue;                    // :35
        float v[9];
        for (int i = 0; i < 3; i++) {                                             // :40-57
            std::string vl;
            getline(vl);
            split(vl, tok);
            if (tok.size() < 3) { rc = MIRT_ERR_INVALID_ARGUMENT; break; }
            for (int c = 0; c < 3; c++) v[3 * i + c] = (float)std::atof(tok[c].c_str());
        }
        if (rc) break;
        if (tris15 && n < max_tris) {
            float *t = tris15 + (size_t)15 * n;
            for (int c = 0; c < 9; c++) t[c] = v[c] * -scale;                     // :66-76 (x, z, y of v0, v1, v2: order is irrelevant)
            t[12] = colour3[0]; t[13] = colour3[1]; t[14] = colour3[2];           // :22
            compute_normal(t);                                                    // :78
        }
        n++;
    }
    std::fclose(f);
    return rc ? rc : n;
}

// The per-frame constants of the cull step (rasteriser.cpp:385-402); host libm for acosf / tanf, as the reference.
void mirt::cull_setup(const mirt_view *view, int flags, CullParams *cp)
{
    memcpy(cp->cam, view->pos, 12);
    memcpy(cp->rot, view->rot, 36);
    const v3 cam = ld3(view->pos);
    const v3 fVec = normalize3(vec_mul_mat3(V3(0, 0, 1.0f), view->rot));            // :385
    const float nearz = cam.z + fVec.z * 0.1f, farz = cam.z + fVec.z * 15.0f;       // :386
    const float w = (float)view->width, h = (float)view->height;
    const v3 t = V3(0.0f, -h / 2.0f, view->focal), b = V3(0.0f, h / 2.0f, view->focal);
    const float cy = dot3(t, b) / (length3(t) * length3(b));                        // :394
    const float rfovy = acosf(cy);                                                  // :395
    const float aspect = w / h;
    memset(cp->tr, 0, sizeof cp->tr);
    cp->tr[0] = (1.0f / tanf(rfovy / 2.0f)) / aspect;                               // transform[0][0] :398
    cp->tr[5] = (1.0f / tanf(rfovy / 2.0f));                                        // transform[1][1] :399
    cp->tr[10] = farz / (farz - nearz);                                             // transform[2][2] :400
    cp->tr[14] = 1.0f;                                                              // transform[3][2] :401-402
    cp->flags = flags;
}

// The cull step of the rasteriser's Update(), rasteriser.cpp:385-447, InCuboid :451-458.
extern "C" int mirt_cull(const float *tris15, int n, const mirt_view *view, int flags, uint8_t *culled)
{
    if (!tris15 || !view || !culled || n < 0) return MIRT_ERR_INVALID_ARGUMENT;
    CullParams cp;
    cull_setup(view, flags, &cp);
    for (int i = 0; i < n; i++) culled[i] = cull_one(tris15 + (size_t)15 * i, cp);
    return MIRT_OK;
}
