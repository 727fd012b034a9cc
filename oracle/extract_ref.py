#!/usr/bin/env python3
"""oracle/extract_ref.py -- TEST INFRASTRUCTURE ONLY.

Copies the SDL-free line ranges of the reference's two render translation units, verbatim, into
git-ignored oracle/_ref/*.inc so that oracle/ref_rt.cpp and oracle/ref_raster.cpp can compile the
reference's OWN text of the hot-path functions against the GLM it vendors -- without SDL (absent in this
image) and without any stand-in for it.  Nothing is retyped: every output line is a line of the reference,
preceded by one `#line` directive per range so that compiler diagnostics point back into the reference.
The .inc files are build products (never committed); the reference itself is only read.

    python3 oracle/extract_ref.py /root/reference oracle/_ref

What is deliberately NOT extracted (needs an SDL type or call, so it stays pinned by SURVEY Appendix C
only): main(), Update()'s keyboard handling and screen clear, CalculateDOF()'s lock/PutPixelSDL/update
calls, SDLauxiliary.h (PutPixelSDL, InitializeSDL), the signature line of DrawLineSDL (its first
parameter is an SDL_Surface*), the `SDL_Surface* screen;` globals.
"""
import os
import sys

# (output file, source file, [(first line, last line, what)]) -- 1-based inclusive line numbers
RANGES = [
    ("rt_globals.inc", "raytracer/Source/raytracer.cpp", [
        (22, 24, "using-declarations"),
        (28, 75, "globals: triangles, render settings, lights, key states, SCREEN_WIDTH/HEIGHT (REALTIME: 150), camera"),
        (77, 98, "globals: indirectLight, randomPositions, focalDistances, pixelColours, struct Intersection, closestIntersections"),
        (103, 112, "function prototypes"),
    ]),
    ("rt_init_intersections.inc", "raytracer/Source/raytracer.cpp", [(152, 162, "main(): closestIntersections fill, cameraRot[1][1] = 1")]),
    ("rt_addlight.inc", "raytracer/Source/raytracer.cpp", [(180, 193, "AddLight")]),
    ("rt_closest.inc", "raytracer/Source/raytracer.cpp", [(202, 257, "ClosestIntersection")]),
    ("rt_random.inc", "raytracer/Source/raytracer.cpp", [(260, 263, "RandomNumber")]),
    ("rt_direct.inc", "raytracer/Source/raytracer.cpp", [(265, 327, "DirectLight")]),
    ("rt_reset.inc", "raytracer/Source/raytracer.cpp", [(335, 339, "Update(): reset of the intersection distances")]),
    ("rt_camera.inc", "raytracer/Source/raytracer.cpp", [(377, 382, "Update(): cameraRot from yaw")]),
    ("rt_draw.inc", "raytracer/Source/raytracer.cpp", [(547, 603, "Draw() up to, not including, its CalculateDOF() call")]),
    ("rt_dof_loop.inc", "raytracer/Source/raytracer.cpp", [(613, 645, "CalculateDOF(): the blur loops up to, not including, PutPixelSDL")]),

    ("ra_globals.inc", "rasteriser/Source/rasteriser.cpp", [
        (9, 15, "using-declarations"),
        (22, 32, "globals: settings"),
        (34, 80, "globals: screen size, camera, lights, depthBuffer, triangles, DOF containers, clip volume"),
    ]),
    ("ra_cull.inc", "rasteriser/Source/rasteriser.cpp", [(377, 447, "Update(): body of `if (isUpdated)` -- cameraRot from yaw and the cull step")]),
    ("ra_incuboid.inc", "rasteriser/Source/rasteriser.cpp", [(451, 458, "InCuboid")]),
    ("ra_draw_loop.inc", "rasteriser/Source/rasteriser.cpp", [(466, 479, "Draw(): the triangle loop")]),
    ("ra_dof_loop.inc", "rasteriser/Source/rasteriser.cpp", [(486, 518, "CalculateDOF(): the blur loops up to, not including, PutPixelSDL")]),
    ("ra_vertex_shader.inc", "rasteriser/Source/rasteriser.cpp", [(532, 546, "VertexShader")]),
    ("ra_pixel_shader.inc", "rasteriser/Source/rasteriser.cpp", [(549, 589, "PixelShader")]),
    ("ra_drawline_body.inc", "rasteriser/Source/rasteriser.cpp", [(593, 612, "DrawLineSDL: body (the signature line names an SDL type)")]),
    ("ra_interpolate.inc", "rasteriser/Source/rasteriser.cpp", [(615, 637, "Interpolate")]),
    ("ra_bresenham.inc", "rasteriser/Source/rasteriser.cpp", [(639, 672, "Bresenham")]),
    ("ra_polygon_rows.inc", "rasteriser/Source/rasteriser.cpp", [(674, 735, "ComputePolygonRows")]),
    ("ra_draw_rows.inc", "rasteriser/Source/rasteriser.cpp", [(738, 753, "DrawRows")]),
    ("ra_draw_polygon.inc", "rasteriser/Source/rasteriser.cpp", [(755, 768, "DrawPolygon")]),
]

# Tripwires: the first line of a few ranges must still be what this recipe was written against.
EXPECT = {
    ("raytracer/Source/raytracer.cpp", 202): "bool ClosestIntersection(",
    ("raytracer/Source/raytracer.cpp", 265): "vec3 DirectLight(",
    ("raytracer/Source/raytracer.cpp", 547): "void Draw()",
    ("raytracer/Source/raytracer.cpp", 76): "SDL_Surface* screen;",
    ("rasteriser/Source/rasteriser.cpp", 33): "SDL_Surface* screen;",
    ("rasteriser/Source/rasteriser.cpp", 532): "void VertexShader(",
    ("rasteriser/Source/rasteriser.cpp", 592): "void DrawLineSDL( SDL_Surface*",
    ("rasteriser/Source/rasteriser.cpp", 674): "void ComputePolygonRows(",
    ("rasteriser/Source/rasteriser.cpp", 755): "void DrawPolygon(",
}


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    cache = {}
    for (path, line), text in EXPECT.items():
        lines = cache.setdefault(path, open(os.path.join(ref, path), encoding="latin-1").read().split("\n"))
        if not lines[line - 1].lstrip().startswith(text):
            sys.exit("extract_ref: %s:%d is %r, expected %r -- the reference changed, re-derive the ranges" % (path, line, lines[line - 1], text))
    for name, path, ranges in RANGES:
        lines = cache[path]
        with open(os.path.join(out, name), "w", encoding="latin-1") as f:
            for first, last, what in ranges:
                for forbidden in lines[first - 1:last]:
                    if "SDL" in forbidden.split("//")[0] and "DrawLineSDL" not in forbidden:
                        sys.exit("extract_ref: %s:%d-%d (%s) contains an SDL token: %r" % (path, first, last, what, forbidden))
                f.write('#line %d "%s"\n' % (first, os.path.join(ref, path)))
                f.write("\n".join(lines[first - 1:last]) + "\n")
    print("extracted %d ranges of the reference into %s" % (sum(len(r[2]) for r in RANGES), out))


if __name__ == "__main__":
    main()
