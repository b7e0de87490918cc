// xm_sell.h — "sliced ELL over virtual rows" storage of a 3x3-block sparse Q for the large-n Q*W product (xm_sell.hip).
//
// Why a second sparse layout (the reference has none: its Q is dense, Dense/matmul.h:42-87): with 3x3-block CSR
// and one 16-lane group per camera row the 100k-camera product is bound twice over (profiles/r01_pmc_bsr3_100k.json) —
// (i) fabric traffic 2.24x the algorithmic bytes, because the gathered operand W (7.2 MB at o = 3) does not fit the 4 MB L2 of
// an XCD and every miss pulls a 128-byte line for a 72-byte record, and (ii) the load path: ~11 load instructions per 64 blocks
// most of which touch many sectors, plus ~290 VALU lane-instructions per block of index arithmetic, LDS staging and reductions
// around the 27 FMAs that matter.  This layout removes both:
//   * COLUMN SLABS PER XCD.  The cameras (columns) are cut into S slabs (S = 1, 2, 4, 8); workgroup b runs on XCD b % 8 (observed
//     dispatch rule, used for speed only) and works on slab (b % 8) * S / 8 only, so an XCD gathers from a 7.2 MB / S slice of W
//     that stays resident in its own L2.  Every camera row is cut into per-slab segments ("virtual rows", additionally cut at
//     `lmax` blocks so that a hub camera cannot serialise a lane); a virtual row produces a partial 3 x o result, and a second
//     small kernel adds the partials of a camera in a fixed order and runs the fused epilogue.
//   * ONE LANE PER VIRTUAL ROW, blocks stored lane-interleaved (sliced ELL): the virtual rows of a slab are sorted by length and
//     packed 64 to a slice; step k of a slice holds the k-th block of each of its 64 virtual rows as 9 planes of 64 doubles
//     (two steps interleaved -> every lane loads 16 aligned bytes, every wave instruction covers 1 KiB of contiguous memory).
//     No LDS staging, no cross-lane reduction, no per-block index arithmetic: per block a lane issues its share of 9.5 coalesced
//     loads, gathers its 3 x o rows of W and does the 9*o FMAs.
// Sorting makes the padding negligible (only slices that straddle two length classes carry any).
#pragma once

#include <cstdint>
#include <vector>

#include "xm_solver.h"

namespace xm {

// host-side description (built by sell_build_host; exported through xm_sell_layout for the CPU tests)
struct SellHost {
    int64_t nloc = 0, ncols = 0;
    int S = 1, lmax = 0;
    bool slice_order = true;   // partial result of (slice, lane) stored at slice * 64 + lane
    int64_t nvrows = 0, nslices = 0, nsteps = 0, nparts = 0, nstore = 0;   // nstore: entries of the partial-result array
    std::vector<int64_t> slice_off;   // nslices + 1, in steps; a slice of width w owns steps [off, off + w)
    std::vector<int32_t> slab_start;  // S + 1, in slices
    std::vector<uint8_t> kind;        // nsteps: 0 = first step of a pair, 1 = second step of a pair, 2 = unpaired last step
    std::vector<int64_t> src;         // nsteps * 64: source block of (step, lane) in the CSR arrays, -1 = padding
    std::vector<int32_t> pslot;       // nslices * 64: where (slice, lane) stores its partial result, -1 = padding lane
    std::vector<int32_t> ridx;        // nparts: storage index of the k-th listed partial result (see pptr)
    std::vector<int64_t> pptr;        // nloc + 1: the partial results of camera r are ridx[pptr[r] .. pptr[r+1])
    // distinct 128-byte lines of W the 64 lanes of a step touch, summed over every 4th step: records of 72 bytes (o = 3) and of 120 bytes
    // (o = 4, 5) at their native pitch, and records at the 128-byte pitch (== distinct columns).  Random graphs: 1.4 / 1.9 against 1 line
    // per record; graphs with column locality (banded): ~0.6 / 1 against 1 -- there the padded copy of W loses (xm_sell.h: Wpad16)
    int64_t lines_native72 = 0, lines_native120 = 0, lines_padded = 0;
    std::vector<int64_t> diag_src;    // view-graph codec only (diag_row0 >= 0): CSR position of row r's diagonal block (-1: none); those
                                      // blocks are NOT in the slices (they are d * I, applied by the second launch from one double)
};

// rowptr: nloc + 1 offsets (rowptr[0] may be non-zero: offsets into colidx); colidx: global columns in [0, ncols).
// Throws Error(XM_ERR_ARG) on a malformed description (non-monotone rowptr, column out of range).
// diag_row0 >= 0: local row r is global camera diag_row0 + r and its diagonal block (column diag_row0 + r) is left out of the slices.
void sell_build_host(const int64_t *rowptr, const int32_t *colidx, int64_t nloc, int64_t ncols, int S, int lmax, SellHost &out,
                     int64_t diag_row0 = -1);

// Block codecs of the slices.
//   SELL_CODEC_FULL  9 doubles per block (any 3x3 block): 76 bytes per stored block with its column index.
//   SELL_CODEC_QUAT  view-graph matrices (the north_star workload: Q = sum_e w_e G_e over view-graph edges, off-diagonal blocks
//                    -w_e M_e with M_e a rotation, diagonal blocks (sum of incident weights) * I): an off-diagonal block is stored as
//                    the quaternion of M_e scaled by sqrt(2 w_e) -- 4 doubles, the block is rebuilt in registers as -R(q) (R is
//                    quadratic in q, so the scale carries the weight; 23 flops) -- and a diagonal block as ONE double per camera.
//                    36 bytes per stored block instead of 76: the HBM stream of the product is 0.47x.  w_e = 0 (an edge removed by
//                    the XM^2 filter) is the zero quaternion.
enum { SELL_CODEC_FULL = 0, SELL_CODEC_QUAT = 1 };
constexpr int64_t kSellMinBlocks = 2200000;   // stored blocks per GPU from which a context builds the sliced-ELL copy by itself (measured cross-over, xm_solver.hip)

struct SellArgs {   // what the kernels see
    const int64_t *slice_off;
    const int32_t *slab_start;
    const int32_t *cols;    // step unit = 64 ints: pair [lane][2] over two units, single [lane]
    const double *blk;      // step unit = 64 * NQ doubles (NQ = 9 | 4 by codec): pair [e][lane][2] over two units, single [e][lane]
    const double *diag;     // quaternion codec: diagonal scalar per local camera (else nullptr)
    int64_t row0;           // global camera index of local row 0 (the diagonal term reads W at row0 + cam)
    const int32_t *pslot;
    const int64_t *pptr;
    const int32_t *ridx;
    int S;
    int wstride;           // doubles between camera records in the W the kernel reads (3 * pitch natively; 16 = one cache line per record)
    int coalesced_store;   // 1: partial results stored at slice * 64 + lane AND written as one contiguous run per slice (through LDS)
    int64_t nt_off;        // slices whose stream offset (slice_off, in step units) is >= this are read non-temporally, the others with the default policy
};

class SellMatrix {
public:
    int64_t ncols() const { return ncols_; }
    // does a copy of W at the 128-byte record pitch shorten the gather of rank o?  (fewer lines per step than at the native pitch, 10 % margin)
    bool padded_pays(int o) const { return o >= 3 && 3 * pitch_of(o) <= 16 && lines_padded_ * 11 < (3 * pitch_of(o) <= 9 ? lines72_ : lines120_) * 10; }
    // blocks: host, 9 doubles per block (row-major 3x3), indexed like colidx
    // codec SELL_CODEC_QUAT: throws Error(XM_ERR_ARG) unless every off-diagonal block is -w * rotation and every diagonal block d * I
    // (relative 1e-9); row0 = global camera index of local row 0
    SellMatrix(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t ncols, int S, int lmax,
               hipStream_t st, int codec = SELL_CODEC_FULL, int64_t row0 = 0);
    int codec() const { return codec_; }
    int64_t stream_bytes() const;   // bytes of block + index stream one product reads
    int64_t nt_off(int o, int64_t nloc) const;   // SellArgs.nt_off of a rank-o product over nloc cameras (xm_sell.hip)
    SellArgs args() const;
    void refill(const int32_t *d_colidx, const double *d_blocks, hipStream_t st);   // values changed on the device (XM^2 re-weighting)
    double *parts(int o);
    int wstride(int o) const;
    int reduce_gw(int o) const;                 // lanes per camera in the second launch (4 | 16)
    int reduce_grid(int o, int nloc) const;     // workgroups of the second launch == per-workgroup partial sums per epilogue slot          // partial-result buffer for rank o (grow-only)
    int grid() const { return grid_; }
    int64_t nloc() const { return nloc_; }
    int64_t nparts() const { return nparts_; }
    int64_t nsteps() const { return nsteps_; }
    int S() const { return S_; }

private:
    int64_t nloc_ = 0, nparts_ = 0, nsteps_ = 0, nslices_ = 0;
    int S_ = 1, grid_ = 0;
    int64_t ncols_ = 0, max_list_ = 0;   // max_list_: most partial results of one camera
    int64_t lines72_ = 0, lines120_ = 0, lines_padded_ = 0;   // SellHost::lines_*
    bool coalesced_ = false;   // partial results written as one contiguous run per slice (slice-order slots)
    DevBuf<int64_t> slice_off_, pptr_;
    DevBuf<int32_t> slab_start_, cols_, pslot_, ridx_;
    DevBuf<double> blk_, parts_;
    DevBuf<int64_t> src_;
    DevBuf<uint8_t> kind_;
    int64_t b0_ = 0;
    int parts_o_ = 0;
    int codec_ = SELL_CODEC_FULL;
    int64_t row0_ = 0;
    DevBuf<double> diag_;
    DevBuf<int64_t> diag_src_;
};

// product = two launches: partial results per virtual row, then per-camera sum + fused epilogue (same CamArgs contract and
// per-workgroup partial sums of the second launch, grid SellMatrix::reduce_grid(o, nloc)).  gm: 0 = each lane loads its own record of W,
// 1 = records fetched element-per-lane and transposed through LDS.
// host check of the view-graph structure the quaternion codec relies on (O(nb)); throws Error(XM_ERR_ARG)
void check_viewgraph_blocks(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t nloc, int64_t row0);
// Wpad16 (optional): the same W at a record pitch of 16 doubles (one 128-byte line per camera; o = 3..5).  The gather then touches one
// line per record instead of 1.4 (o = 3) / 1.9 (o = 4, 5): main launch 72.8 -> 63.6 us (o = 3), 89.2 -> 79.4 us (o = 4) at 100 k
// cameras (profiles/r04_trace_sell_pitch16.txt).  The solver's tCG writes that copy from the kernels that produce W (tcg_init / cg_step).
void launch_qw_sell(int o, int epi, SellMatrix &m, const double *W, double alpha, const CamArgs &a, int gm, hipStream_t st,
                    const double *Wpad16 = nullptr);
bool sell_supports(int o);
void sell_quat_roundtrip(const double block[9], double quat[4], double rebuilt[9]);   // host: the codec's two maps

}  // namespace xm
