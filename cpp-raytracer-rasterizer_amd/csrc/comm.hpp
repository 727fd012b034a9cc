// comm.hpp -- band partition of a sharded frame and the gather of the bands on the root (comm.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace mirt {

constexpr int COMM_ID_BYTES = 128;      // == NCCL_UNIQUE_ID_BYTES

// rows [y0, y1) of `rank` when `height` rows are split into `world` contiguous bands (sizes differ by at most one row)
void band_of(int rank, int world, int height, int *y0, int *y1);

// The rows of a rank as SEGMENTS of the frame, for either partition:
//   strip_rows == 0   one contiguous band per rank (band_of): one binning pass and one launch chain per rank and frame -- what the
//                     binned ray tracer wants, its per-frame cost having a part that does not shrink with the rows rendered;
//   strip_rows  > 0   interleaved strips of that many rows, strip s to rank s % world -- every rank samples the whole height of the
//                     frame, which evens out scenes whose cost is concentrated in some rows (the reference's schedule(auto) over
//                     rows, raytracer.cpp:557, at the granularity a GPU launch needs).
// A rank's band buffer holds its segments of one view back to back, views one after the other.
//   bounds != NULL    contiguous bands with explicit boundaries: rank r renders rows [bounds[r], bounds[r + 1]) -- the weighted
//                     partition (part_weighted_bounds), bands of equal estimated cost; strip_rows is ignored.
int part_segments(int rank, int world, int height, int strip_rows, const int *bounds = nullptr);                          // how many segments the rank has
void part_segment(int rank, int world, int height, int strip_rows, int k, int *y0, int *y1, const int *bounds = nullptr);  // its k-th: rows [y0, y1)
int part_rows(int rank, int world, int height, int strip_rows, const int *bounds = nullptr);                              // its rows in total
// world + 1 boundaries (multiples of 8 rows) of bands of equal estimated cost; pure integer arithmetic (comm.cpp)
void part_weighted_bounds(const uint32_t *hist, int hist_rows, int shift, int width, int height, int world, unsigned tile_weight, int *bounds);

struct Comm;
// rank 0 creates the id (ncclGetUniqueId; a file-name prefix for the shm transport) and hands it to the other ranks
bool comm_create_id(void *id128);
Comm *comm_init(const void *id128, int rank, int world);      // collective; NULL on failure (comm_error(NULL))
void comm_destroy(Comm *c);
const char *comm_error(const Comm *c);
int comm_rank(const Comm *c);
int comm_world(const Comm *c);

// The messages of one gather as offsets: on the root, `bytes` from rank `peer` land at root_offset of its frame buffer
// (nviews frames of height * width * 4 bytes); on rank `peer`, they are the `bytes` at band_offset of its band buffer (nviews
// bands of its rows).  Returns the number of pieces written (at most max_pieces); pure arithmetic.
struct BandPiece { size_t root_offset, band_offset, bytes; int peer; };
int band_gather_plan(int world, int root, int width, int height, int nviews, BandPiece *out, int max_pieces);
// ... for either partition: one piece per (rank, view, segment)
int part_gather_plan(int world, int root, int width, int height, int nviews, int strip_rows, BandPiece *out, int max_pieces, const int *bounds = nullptr);

// One message of the gather.  On the root: `bytes` from rank `peer` land at `ptr`; elsewhere: `bytes` at `ptr` go to the root.
struct GatherPiece { void *ptr; size_t bytes; int peer; };
// All pieces in one group on `stream` (RCCL: asynchronous, stream-ordered; shm: synchronous).
bool comm_gather_bands(Comm *c, int root, const GatherPiece *pieces, int npieces, hipStream_t stream);

// Link check: `bytes` of a pattern travel from this rank to ITSELF through the group's transport (one send + one receive in a
// group, as in a gather) and are compared.  Works with any world size, so it also covers the transport on a one-GPU box.
bool comm_selfcheck(Comm *c, size_t bytes, hipStream_t stream);

}  // namespace mirt
