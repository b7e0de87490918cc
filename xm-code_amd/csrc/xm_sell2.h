// xm_sell2.h — "chunk-tiled sliced ELL": the block-sparse Q*W product with its fused epilogue in ONE launch (xm_sell2.hip).
//
// Replaces, for view-graph-sparse Q, the product the reference runs as cublasDgemm on a dense matrix (Dense/matmul.h:42-87, call sites
// trustregion.h:165,187,237,553, checkeig.h:182).  Second generation of xm_sell.h, which needs two launches (partial results per virtual
// row, then a per-camera sum + epilogue kernel) and a table look-up per partial result: at 100 k cameras the second launch was 15-18 us
// of an 88 us product and the partial results made a 50 MB round trip behind a launch boundary (profiles/r03_pmc_sell_quat_100k.txt).
//
//   * CHUNKS AND TILES.  The local cameras are cut into chunks of 64 (one lane each when the results are summed).  The blocks of a chunk
//     that fall into one COLUMN SLAB (slabs as in xm_sell.h: a slab of W stays in the L2 of the XCDs that serve it) form one SLICE, the
//     work of one wavefront; a slice produces a TILE = the 64 x (3 x o) partial result of its chunk, and a chunk normally owns S tiles.
//   * LANES BALANCED BY CONSTRUCTION, NOT BY SORTING.  The blocks of a slice are laid out row after row and dealt to the 64 lanes in
//     equal contiguous runs of K = ceil(B / 64) blocks, whatever the row boundaries are (a lane may finish one camera's row and start the
//     next; a long row runs over several lanes).  Every lane does the same number of steps, the padding is < 64 blocks per slice, and no
//     global sort by row length is needed -- which is what ties a slice to ONE chunk and makes the hand-off below local.  A lane keeps one
//     accumulator; a flag in the column word (bit 31) says "the row ends after this block": the sum goes to the row's slot in LDS (or stays
//     in registers when it is the lane's first run, which may be the continuation of the previous lane's row) and the accumulator restarts.
//     Afterwards lane r adds, in block order, what belongs to row r: its LDS slot and the first runs of the lanes its row continued into.
//     A (chunk, slab) list longer than 64 * kmax blocks (hub cameras) is cut into several slices, each with its own tile.
//   * ONE LAUNCH.  A slice stores its tile with write-through (agent-scope) stores, drains them (s_waitcnt) and takes a ticket on the
//     chunk's arrival counter.  The wavefront that arrives LAST for a chunk adds the chunk's tiles in tile order -- fixed, so the result
//     does not depend on who was last: bit-reproducible -- and runs the fused epilogue (xm_device.h semantics) with one lane per camera;
//     per-chunk partial sums replace the per-workgroup partial sums of the second launch.  No cache-wide fence anywhere (MI355X_MICROARCH:
//     "sc1 stores AND sc1 loads"), no atomics on data.
// Block codecs, slab rule, gather modes and the odd-pitch layout of W are those of xm_sell.h.
#pragma once

#include <cstdint>
#include <vector>

#include "xm_sell.h"

namespace xm {

constexpr int kSell2Chunk = 64;            // cameras per chunk == lanes
constexpr int64_t kSell2MaxCols = 1 << 24; // column word = column (24 bits) | row inside the chunk (6 bits) << 24 | row-end flag << 31

// host-side description (built by sell2_build_host; exported through xm_sell2_layout for the CPU tests)
struct Sell2Host {
    int64_t nloc = 0, ncols = 0;
    int S = 1, kmax = 0;
    int64_t nchunks = 0, nslices = 0, nsteps = 0, ntiles = 0;
    std::vector<int64_t> slice_off;    // nslices + 1, in steps; slice c owns steps [off, off + K)
    std::vector<int32_t> slab_start;   // S + 1, in slices (the slices of a slab are in chunk order)
    std::vector<int32_t> slice_chunk;  // nslices
    std::vector<int32_t> slice_tile;   // nslices: index of the tile the slice writes (tiles of a chunk are consecutive)
    std::vector<int32_t> tile_ptr;     // nchunks + 1: chunk k owns tiles [tile_ptr[k], tile_ptr[k+1]) and expects that many arrivals
    std::vector<uint8_t> kind;         // nsteps: 0 = first step of a pair, 1 = second step of a pair, 2 = unpaired last step
    std::vector<int64_t> src;          // nsteps * 64, entry (step, lane) = block lane * K + (step - off) of the slice: CSR position in bits 0..47,
                                       // row inside the chunk in bits 48..53, row-end flag in bit 54; -1 = padding
    std::vector<int32_t> lane_meta;    // nslices * 64, entry (slice, lane): bits 0..5 la, bits 6..12 nA -- row `lane` of the chunk adds the first
                                       // runs of lanes la .. la + nA - 1 to its LDS slot; bit 13 hasZ, bits 14..19 zrow -- lane `lane` ends inside a
                                       // row (after at least one row end): its last run goes to the LDS slot of row zrow
    std::vector<int64_t> diag_src;     // view-graph codec only: CSR position of row r's diagonal block (-1: none)
};

// rowptr: nloc + 1 offsets (rowptr[0] may be non-zero); colidx: global columns in [0, ncols), ncols < kSell2MaxCols.
// diag_row0 >= 0: local row r is global camera diag_row0 + r and its diagonal block is left out of the slices (view-graph codec).
void sell2_build_host(const int64_t *rowptr, const int32_t *colidx, int64_t nloc, int64_t ncols, int S, int kmax, Sell2Host &out,
                      int64_t diag_row0 = -1);

struct Sell2Args {   // what the kernel sees
    const int64_t *slice_off;
    const int32_t *slab_start;
    const int32_t *slice_chunk, *slice_tile, *tile_ptr, *lane_meta;
    const int32_t *cols;    // as xm_sell.h: pair [lane][2] over two step units, single [lane]
    const double *blk;      // pair [e][lane][2], single [e][lane]; e < 9 | 4 by codec
    const double *diag;     // quaternion codec: diagonal scalar per local camera (else nullptr)
    int64_t row0;
    double *tiles;          // ntiles x (3 x o) x 64: per tile 3 x o planes of 64 lane values
    unsigned int *arrived;  // nchunks arrival counters, zero between launches
    int nchunks;
    int S;
    int wstride;
};

class Sell2Matrix {
public:
    Sell2Matrix(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t ncols, int S, int kmax, hipStream_t st,
                int codec = SELL_CODEC_FULL, int64_t row0 = 0);
    int codec() const { return codec_; }
    int64_t stream_bytes() const;   // bytes of block + index stream one product reads
    Sell2Args args(int o);          // sizes the tile buffer for rank o (grow-only)
    void refill(const int32_t *d_colidx, const double *d_blocks, hipStream_t st);   // values changed on the device (XM^2 re-weighting)
    int grid() const { return grid_; }
    int nchunks() const { return (int)nchunks_; }   // == per-chunk partial sums per epilogue slot
    int64_t nloc() const { return nloc_; }
    int64_t nsteps() const { return nsteps_; }
    int64_t ntiles() const { return ntiles_; }
    int S() const { return S_; }
    static bool supports(int o, int64_t ncols) { return (o == 1 || (o >= 3 && o <= 5)) && ncols < kSell2MaxCols; }

private:
    int64_t nloc_ = 0, ncols_ = 0, nsteps_ = 0, nslices_ = 0, nchunks_ = 0, ntiles_ = 0;
    int S_ = 1, grid_ = 0, codec_ = SELL_CODEC_FULL, tiles_o_ = 0;
    int64_t b0_ = 0, row0_ = 0;
    DevBuf<int64_t> slice_off_, src_, diag_src_;
    DevBuf<int32_t> slab_start_, slice_chunk_, slice_tile_, tile_ptr_, lane_meta_, cols_;
    DevBuf<uint8_t> kind_;
    DevBuf<double> blk_, tiles_, diag_;
    DevBuf<unsigned int> arrived_;
};

// one launch: product + per-camera sum + fused epilogue; CamArgs contract of the other Q*W kernels with per-CHUNK partial sums
// (a.partials[slot * nchunks + chunk]).  gm: 0 = each lane loads its own record of W, 1 = records fetched element-per-lane and
// transposed through LDS.  pipe: 0 = single-buffered, 1 = block loads one pair ahead, -1 = default for (o, codec).
void launch_qw_sell2(int o, int epi, Sell2Matrix &m, const double *W, double alpha, const CamArgs &a, int gm, int pipe, hipStream_t st);

}  // namespace xm
