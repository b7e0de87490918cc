// xm_capi.hip — extern "C" boundary (include/xm_amd.h).  No exceptions cross it.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <mutex>
#include <vector>

#include "xm_schur.h"
#include "xm_sell.h"
#include "xm_symw.h"
#include "xm_solver.h"

struct xm_ctx {
    std::unique_ptr<xm::Context> impl;   // one GPU (or one rank of a multi-process run)
    std::unique_ptr<xm::Team> team;      // n_gpus > 1: single-process multi-GPU
};

namespace {
thread_local std::string g_err;
int fail(const xm::Error &e) { g_err = e.what(); return e.code; }
int fail(const std::exception &e) { g_err = e.what(); return XM_ERR_HIP; }

#define XM_TRY try {
#define XM_CATCH                                                       \
    }                                                                  \
    catch (const xm::Error &e) { return fail(e); }                     \
    catch (const std::bad_alloc &) { g_err = "out of host memory"; return XM_ERR_NOMEM; } \
    catch (const std::exception &e) { return fail(e); }

// Revision-3 structs start with the caller's sizeof: copy what the caller has, zero the rest (include/xm_amd.h, XM_ABI_REVISION)
template <class T>
T take_struct(const T *p, const char *what) {
    T out;
    std::memset(&out, 0, sizeof(T));
    if (!p) throw xm::Error(XM_ERR_ARG, std::string(what) + ": null");
    const uint32_t sz = p->struct_size;
    if (sz < 16 || sz > 4096) throw xm::Error(XM_ERR_ARG, std::string(what) + ": struct_size is not set (ABI revision 3: the first field of the struct is sizeof(struct))");
    std::memcpy(&out, p, std::min<size_t>(sz, sizeof(T)));
    out.struct_size = (uint32_t)sizeof(T);
    return out;
}
void give_result(xm_result_t *dst, const xm_result_t &src, uint32_t caller_size) {
    xm_result_t tmp = src;
    tmp.struct_size = caller_size;
    std::memcpy(dst, &tmp, std::min<size_t>(caller_size, sizeof(xm_result_t)));
}

void require_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt < 1)
        throw xm::Error(XM_ERR_HIP, "no HIP device available: the XM solver has no CPU fallback (it needs an MI355X / gfx950 GPU)");
}

// .bin matrix: int32 rows, int32 cols, float64 column-major (XM_main.cu:18-33, utils/io.py:17-54).  The reference's Python I/O
// also knows a variant with two 8-byte header fields (utils/io.py:24-26 `byte = 8`) that its C++ loader cannot read; it is
// accepted here (told apart by the file size), which lifts the int32 limit on rows*cols for large Q.
void read_bin(const std::string &fn, std::vector<double> &d, int64_t &rows, int64_t &cols) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate);
    if (!f) throw xm::Error(XM_ERR_IO, "cannot open file " + fn);
    const int64_t size = (int64_t)f.tellg();
    f.seekg(0);
    unsigned char raw[16] = {0};
    f.read(reinterpret_cast<char *>(raw), std::min<int64_t>(16, size));
    int32_t h4[2];
    int64_t h8[2];
    std::memcpy(h4, raw, 8);
    std::memcpy(h8, raw, 16);
    int64_t skip;
    const bool v1 = size >= 8 && h4[0] >= 0 && h4[1] >= 0 && 8 + 8 * (int64_t)h4[0] * (int64_t)h4[1] == size;
    const bool v2 = !v1 && size >= 16 && h8[0] >= 0 && h8[1] >= 0 && h8[0] < (1LL << 31) && h8[1] < (1LL << 31) &&
                    16 + 8 * h8[0] * h8[1] == size;
    if (v2) { rows = h8[0]; cols = h8[1]; skip = 16; }
    else {
        if (size < 8 || h4[0] < 0 || h4[1] < 0) throw xm::Error(XM_ERR_IO, "bad header in " + fn);
        rows = h4[0]; cols = h4[1]; skip = 8;
        if (8 + 8 * rows * cols > size) throw xm::Error(XM_ERR_IO, "short file " + fn);
    }
    f.clear();
    f.seekg(skip);
    d.resize((size_t)rows * (size_t)cols);
    f.read(reinterpret_cast<char *>(d.data()), (std::streamsize)(d.size() * sizeof(double)));
    if ((size_t)f.gcount() != d.size() * sizeof(double)) throw xm::Error(XM_ERR_IO, "short file " + fn);
}
// header of a .bin matrix without reading the data: rows, cols and the byte offset of element (0, 0)
void peek_bin(const std::string &fn, int64_t &rows, int64_t &cols, int64_t &skip) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate);
    if (!f) throw xm::Error(XM_ERR_IO, "cannot open file " + fn);
    const int64_t size = (int64_t)f.tellg();
    f.seekg(0);
    unsigned char raw[16] = {0};
    f.read(reinterpret_cast<char *>(raw), std::min<int64_t>(16, size));
    int32_t h4[2];
    int64_t h8[2];
    std::memcpy(h4, raw, 8);
    std::memcpy(h8, raw, 16);
    const bool v1 = size >= 8 && h4[0] >= 0 && h4[1] >= 0 && 8 + 8 * (int64_t)h4[0] * (int64_t)h4[1] == size;
    const bool v2 = !v1 && size >= 16 && h8[0] >= 0 && h8[1] >= 0 && h8[0] < (1LL << 31) && h8[1] < (1LL << 31) && 16 + 8 * h8[0] * h8[1] == size;
    if (v2) { rows = h8[0]; cols = h8[1]; skip = 16; }
    else {
        if (size < 8 || h4[0] < 0 || h4[1] < 0) throw xm::Error(XM_ERR_IO, "bad header in " + fn);
        rows = h4[0]; cols = h4[1]; skip = 8;
        if (8 + 8 * rows * cols > size) throw xm::Error(XM_ERR_IO, "short file " + fn);
    }
}
// rows [r0, r0 + nr) of a column-major .bin matrix, all columns -> dst (column-major, leading dimension nr): one contiguous
// piece per column, so a rank of a multi-GPU run reads 1/world of Q.bin instead of all of it
void read_bin_rows(const std::string &fn, int64_t r0, int64_t nr, std::vector<double> &dst, int64_t rows, int64_t cols, int64_t skip) {
    std::ifstream f(fn, std::ios::binary);
    if (!f) throw xm::Error(XM_ERR_IO, "cannot open file " + fn);
    dst.resize((size_t)std::max<int64_t>(nr, 0) * (size_t)cols);
    for (int64_t c = 0; c < cols && nr > 0; ++c) {
        f.seekg(skip + 8 * (c * rows + r0));
        f.read(reinterpret_cast<char *>(dst.data() + (size_t)c * nr), (std::streamsize)(nr * 8));
        if (f.gcount() != (std::streamsize)(nr * 8)) throw xm::Error(XM_ERR_IO, "short file " + fn);
    }
}
void write_bin(const std::string &fn, const double *d, int32_t rows, int32_t cols) {
    std::ofstream f(fn, std::ios::binary);
    if (!f) throw xm::Error(XM_ERR_IO, "cannot write " + fn);
    f.write(reinterpret_cast<const char *>(&rows), 4);
    f.write(reinterpret_cast<const char *>(&cols), 4);
    f.write(reinterpret_cast<const char *>(d), (std::streamsize)((size_t)rows * (size_t)cols * sizeof(double)));
}

int solve_path(const char *dataset_path, unsigned max_rank, double tol, double lam, double max_time, int mode, int *status) {
    if (!dataset_path) throw xm::Error(XM_ERR_ARG, "dataset_path is NULL");
    require_device();
    const std::string base(dataset_path);
    std::vector<double> Q, sini;
    int64_t rows = 0, cols = 0, skip = 0;
    const std::shared_ptr<xm::Comm> cmp = xm::default_comm();
    const xm::Comm &cm = *cmp;
    const bool strip = cm.world > 1;                    // row-partitioned run: this rank needs only the rows of its cameras
    if (strip) peek_bin(base + "/Q.bin", rows, cols, skip);
    else read_bin(base + "/Q.bin", Q, rows, cols);      // XM_main.cu:185
    if (rows != cols || rows % 3 != 0 || rows < 3) throw xm::Error(XM_ERR_IO, "Q.bin must be 3n x 3n");
    const bool verbose = std::getenv("XM_QUIET") == nullptr;
    if (verbose) printf("rows: %lld, cols: %lld\n", (long long)rows, (long long)cols);
    const int64_t n = rows / 3;
    if (mode == XM_MODE_REBUTTLE) {
        int64_t r2, c2;
        read_bin(base + "/s_ini.bin", sini, r2, c2);      // XM_main.cu:42,62
        if ((int64_t)sini.size() < n) throw xm::Error(XM_ERR_IO, "s_ini.bin too short");
        std::vector<double> rini;
        read_bin(base + "/R_ini.bin", rini, r2, c2);      // read like the reference (XM_main.cu:41,61); its content is then
                                                          // overwritten by the identity stack at rank 3 (XM_main.cu:95-103)
    }
    xm_problem_t prob;
    std::memset(&prob, 0, sizeof(prob));
    prob.struct_size = sizeof(prob);
    prob.n = n; prob.storage = XM_STORAGE_DENSE; prob.q = Q.data(); prob.ldq = rows;
    // XM_GPUS=N: the reference's own single-process call on N GPUs of this node (XM_GPU_MAP=1: N virtual ranks on device 0)
    const int n_gpus = (int)std::max(1L, std::min(8L, std::getenv("XM_GPUS") ? std::atol(std::getenv("XM_GPUS")) : 1L));
    const int gpu_map = std::getenv("XM_GPU_MAP") ? std::atoi(std::getenv("XM_GPU_MAP")) : 0;
    if (strip) {
        const int64_t per = xm::equal_range_len(n, cm.world);                    // same partition as xm_partition / Context
        const int64_t c0 = std::min<int64_t>(n, (int64_t)cm.rank * per), c1 = std::min<int64_t>(n, (int64_t)(cm.rank + 1) * per);
        read_bin_rows(base + "/Q.bin", 3 * c0, 3 * (c1 - c0), Q, rows, cols, skip);
        prob.q = Q.data(); prob.ldq = 3 * (c1 - c0); prob.q_row0 = 3 * c0;
        if (c1 == c0) { Q.assign(1, 0.0); prob.q = Q.data(); prob.ldq = 0; prob.q_row0 = 3 * c0; }
    }
    std::unique_ptr<xm::Context> ctx;
    std::unique_ptr<xm::Team> team;
    if (!strip && n_gpus > 1) team.reset(new xm::Team(prob, n_gpus, gpu_map));
    else ctx.reset(new xm::Context(prob));
    std::vector<double>().swap(Q);
    const unsigned rmax = std::max(3u, max_rank);
    std::vector<double> R((size_t)rows * (rmax + 1), 0.0), s((size_t)n, 1.0);
    xm_options_t opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
    opt.max_rank = max_rank; opt.tol = tol; opt.lam = lam; opt.max_time = max_time; opt.mode = mode;
    opt.flags = verbose ? XM_FLAG_VERBOSE : 0;
    opt.s_ini = sini.empty() ? nullptr : sini.data();
    if (const char *e = std::getenv("XM_RETRACTION")) opt.retraction = (*e == 'p' || *e == 'P' || *e == '1') ? XM_RETRACT_POLAR : XM_RETRACT_QR;
    xm_result_t res;
    std::memset(&res, 0, sizeof(res));
    res.R = R.data(); res.s = s.data();
    if (team) team->solve(opt, res); else ctx->solve(opt, res);
    if (cm.rank == 0) {
        write_bin(base + "/R.bin", R.data(), (int32_t)rows, res.rank);   // XM_main.cu:284-294
        if (verbose) printf("saved R\n");
        write_bin(base + "/s.bin", s.data(), (int32_t)n, 1);             // XM_main.cu:298-305
    }
    if (status) *status = res.status;
    return XM_OK;
}
}  // namespace

extern "C" {

const char *xm_last_error(void) { return g_err.c_str(); }
const char *xm_version(void) { return "xm-amd 0.5 (gfx950)"; }
int xm_abi_revision(void) { return XM_ABI_REVISION; }

int xm_solve(const char *p, unsigned int max_rank, double tol, double lam, double max_time) {
    XM_TRY return solve_path(p, max_rank, tol, lam, max_time, XM_MODE_SOLVE, nullptr); XM_CATCH
}
int xm_solve_rank3(const char *p, unsigned int max_rank, double tol, double lam, double max_time) {
    XM_TRY return solve_path(p, max_rank, tol, lam, max_time, XM_MODE_RANK3, nullptr); XM_CATCH
}
int xm_solve_rebuttle(const char *p, unsigned int max_rank, double tol, double lam, double max_time, int *status) {
    XM_TRY return solve_path(p, max_rank, tol, lam, max_time, XM_MODE_REBUTTLE, status); XM_CATCH
}

int xm_ctx_create(const xm_problem_t *prob, xm_ctx_t **out) {
    XM_TRY
    if (!prob || !out) throw xm::Error(XM_ERR_ARG, "null argument");
    require_device();
    const xm_problem_t pr = take_struct(prob, "xm_problem_t");
    auto *c = new xm_ctx;
    try {
        if (pr.n_gpus > 1) {
            if (xm::default_comm()->world > 1) throw xm::Error(XM_ERR_ARG, "n_gpus > 1 inside a multi-process run (xm_comm_init): use one or the other");
            c->team.reset(new xm::Team(pr, pr.n_gpus, pr.gpu_map));
        } else {
            c->impl.reset(new xm::Context(pr));
        }
    } catch (...) { delete c; throw; }
    *out = c;
    return XM_OK;
    XM_CATCH
}
int xm_ctx_solve(xm_ctx_t *ctx, const xm_options_t *opt, xm_result_t *res) {
    XM_TRY
    if (!ctx || !opt || !res) throw xm::Error(XM_ERR_ARG, "null argument");
    const xm_options_t op = take_struct(opt, "xm_options_t");
    xm_result_t rs = take_struct(res, "xm_result_t");
    const uint32_t caller = res->struct_size;
    if (ctx->team) ctx->team->solve(op, rs); else ctx->impl->solve(op, rs);
    give_result(res, rs, caller);
    return XM_OK;
    XM_CATCH
}
void xm_ctx_destroy(xm_ctx_t *ctx) { delete ctx; }
int xm_ctx_qw(xm_ctx_t *ctx, int o, const double *W, double *out, double alpha) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    if (!ctx->impl) throw xm::Error(XM_ERR_ARG, "xm_ctx_qw: single-GPU contexts only");
    ctx->impl->apply(o, W, out, alpha);
    return XM_OK;
    XM_CATCH
}
int xm_ctx_attach_edges(xm_ctx_t *ctx, int64_t ne, const int32_t *ei, const int32_t *ej, const double *M) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    if (ctx->team) ctx->team->attach_edges(ne, ei, ej, M); else ctx->impl->attach_edges(ne, ei, ej, M);
    return XM_OK;
    XM_CATCH
}
int xm_ctx_edge_residuals(xm_ctx_t *ctx, double *res) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    if (ctx->team) ctx->team->edge_residuals(res); else ctx->impl->edge_residuals(res);
    return XM_OK;
    XM_CATCH
}
int xm_ctx_recover_tp(xm_ctx_t *ctx, const double *rot, const double *scale, double *t, double *p) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    if (!ctx->impl) throw xm::Error(XM_ERR_ARG, "xm_ctx_recover_tp: single-GPU contexts only");
    ctx->impl->recover_tp(rot, scale, t, p);
    return XM_OK;
    XM_CATCH
}
int xm_ctx_schur_info(xm_ctx_t *ctx, int *uses_cg, int64_t stats[3], double *last_relres) {
    XM_TRY
    if (!ctx || !ctx->impl || !uses_cg) throw xm::Error(XM_ERR_ARG, "xm_ctx_schur_info: single-GPU context and a non-null output needed");
    int64_t st[3] = {0, 0, 0};
    double rr = 0.0;
    *uses_cg = ctx->impl->schur_info(st, &rr) ? 1 : 0;
    if (stats) { stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; }
    if (last_relres) *last_relres = rr;
    return XM_OK;
    XM_CATCH
}
int xm_ctx_set_edge_weights(xm_ctx_t *ctx, const double *w) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    if (ctx->team) ctx->team->set_edge_weights(w); else ctx->impl->set_edge_weights(w);
    return XM_OK;
    XM_CATCH
}
int64_t xm_dense_ld(int64_t n) { return xm::dense_ld(n); }

int xm_dev_count(int *count) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    if (count) *count = c;
    return XM_OK;
}
int xm_dev_alloc(void **ptr, size_t bytes) {
    XM_TRY require_device(); const size_t padded = (bytes ? bytes : 8) + 128;   /* slack: the sector-window gather reads whole 64-byte sectors around a record */
    XM_HIP_CHECK(hipMalloc(ptr, padded)); XM_HIP_CHECK(hipMemset(*ptr, 0, padded)); XM_HIP_CHECK(hipDeviceSynchronize()); return XM_OK; XM_CATCH
}
int xm_dev_free(void *ptr) { XM_TRY XM_HIP_CHECK(hipFree(ptr)); return XM_OK; XM_CATCH }
int xm_dev_h2d(void *dst, const void *src, size_t bytes) { XM_TRY XM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return XM_OK; XM_CATCH }
int xm_dev_d2h(void *dst, const void *src, size_t bytes) { XM_TRY XM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return XM_OK; XM_CATCH }
int xm_dev_sync(void) { XM_TRY XM_HIP_CHECK(hipDeviceSynchronize()); return XM_OK; XM_CATCH }

int xm_dense_upload(const double *q_host, int64_t ldq, int64_t n, double **dq) {
    XM_TRY
    require_device();
    if (!q_host || !dq || ldq < 3 * n) throw xm::Error(XM_ERR_ARG, "bad argument");
    const int64_t ld = xm::dense_ld(n), m = 3 * n;
    double *tmp = nullptr, *out = nullptr;
    XM_HIP_CHECK(hipMalloc((void **)&tmp, (size_t)m * m * sizeof(double)));
    XM_HIP_CHECK(hipMalloc((void **)&out, (size_t)m * ld * sizeof(double)));
    XM_HIP_CHECK(hipMemcpy2D(tmp, (size_t)m * sizeof(double), q_host, (size_t)ldq * sizeof(double), (size_t)m * sizeof(double), (size_t)m,
                             hipMemcpyHostToDevice));
    xm::launch_transpose_pad(tmp, m, m, m, out, ld, nullptr);
    XM_HIP_CHECK(hipDeviceSynchronize());
    XM_HIP_CHECK(hipFree(tmp));
    *dq = out;
    return XM_OK;
    XM_CATCH
}

int xm_dense_from_bsr3(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t n, double **dq) {
    XM_TRY
    require_device();
    if (!rowptr || !colidx || !blocks || !dq || n < 1) throw xm::Error(XM_ERR_ARG, "bad argument");
    const int64_t ld = xm::dense_ld(n), nb = rowptr[n];
    xm::DevBuf<int64_t> rp; xm::DevBuf<int32_t> ci; xm::DevBuf<double> bl;
    rp.alloc((size_t)n + 1, false); ci.alloc((size_t)nb, false); bl.alloc((size_t)nb * 9, false);
    XM_HIP_CHECK(hipMemcpy(rp.p, rowptr, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    XM_HIP_CHECK(hipMemcpy(ci.p, colidx, (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice));
    XM_HIP_CHECK(hipMemcpy(bl.p, blocks, (size_t)nb * 9 * sizeof(double), hipMemcpyHostToDevice));
    double *out = nullptr;
    XM_HIP_CHECK(hipMalloc((void **)&out, (size_t)3 * n * ld * sizeof(double)));
    XM_HIP_CHECK(hipMemset(out, 0, (size_t)3 * n * ld * sizeof(double)));
    xm::launch_dense_from_bsr(rp.p, ci.p, bl.p, n, 0, out, ld, nullptr);
    XM_HIP_CHECK(hipDeviceSynchronize());
    *dq = out;
    return XM_OK;
    XM_CATCH
}

xm::CamArgs plain_args(int64_t n, double *out) {
    xm::CamArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nloc = (int)n;
    a.out = out;
    return a;
}
int xm_qw_dense(const double *dq, int64_t n, int o, const double *dW, double *dOut, double alpha, void *stream) {
    XM_TRY
    xm::launch_qw_dense(o, xm::EPI_PLAIN, dq, xm::dense_ld(n), dW, alpha, plain_args(n, dOut), (hipStream_t)stream);
    return XM_OK;
    XM_CATCH
}
int xm_qw_dense_sym(const double *dq, int64_t n, int o, const double *dW, double *dOut, double alpha, void *stream) {
    XM_TRY
    const int64_t ld = xm::dense_ld(n);
    xm::DevBuf<double> prow, pcol;
    prow.alloc(xm::sym_prow_count((int)n, ld, o));
    pcol.alloc(xm::sym_pcol_count((int)n, ld, o), false);
    xm::launch_qw_sym(o, xm::EPI_PLAIN, dq, ld, dW, alpha, plain_args(n, dOut), prow.p, pcol.p, (hipStream_t)stream);
    XM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return XM_OK;
    XM_CATCH
}
int xm_qw_bsr3(const int64_t *rp, const int32_t *ci, const double *bl, int64_t n, int o, const double *dW, double *dOut, double alpha,
               void *stream) {
    XM_TRY
    xm::launch_qw_bsr3(o, xm::EPI_PLAIN, rp, ci, bl, dW, alpha, plain_args(n, dOut), (hipStream_t)stream);
    return XM_OK;
    XM_CATCH
}
int xm_retract(int64_t n, int o, const double *dR, const double *ds, const double *dD, const double *dds, double t, double *dRout,
               double *dsout, void *stream) {
    XM_TRY
    xm::launch_retract(o, (int)n, 0, dR, ds, dD, dds, t, dRout, dsout, nullptr, (hipStream_t)stream);
    return XM_OK;
    XM_CATCH
}
int xm_retract_polar(int64_t n, int o, const double *dR, const double *ds, const double *dD, const double *dds, double t, double *dRout,
                     double *dsout, void *stream) {
    XM_TRY
    xm::launch_retract(o, (int)n, 0, dR, ds, dD, dds, t, dRout, dsout, nullptr, (hipStream_t)stream, 1);
    return XM_OK;
    XM_CATCH
}



// inverse of a symmetric positive definite matrix on the device (xm_dense_la.hip; set-up step of the matrix-free storage)
int xm_spd_inverse(int64_t n, double *A) {
    XM_TRY
    require_device();
    if (n < 1 || n > 46000 || !A) throw xm::Error(XM_ERR_ARG, "bad argument");
    xm::DevBuf<double> a, x;
    a.alloc((size_t)n * n, false); x.alloc((size_t)n * n, false);
    XM_HIP_CHECK(hipMemcpy(a.p, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    if (!xm::spd_inverse_device((int)n, a.p, x.p, nullptr)) throw xm::Error(XM_ERR_ARG, "matrix is not positive definite");
    xm::spd_inverse_layout((int)n, x.p, a.p, n, nullptr);   // full symmetric matrix from the computed lower triangle (a is free now)
    XM_HIP_CHECK(hipDeviceSynchronize());
    XM_HIP_CHECK(hipMemcpy(A, a.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost));
    return XM_OK;
    XM_CATCH
}

int xm_symv_plan(int64_t n, int32_t plan[4]) {
    XM_TRY
    if (n < 1 || !plan) throw xm::Error(XM_ERR_ARG, "bad argument");
    int out[4];
    xm::symv_plan_get((int)n, xm::dense_ld(n), out);   // host only: no device needed
    for (int i = 0; i < 4; ++i) plan[i] = out[i];
    return XM_OK;
    XM_CATCH
}

// ---- sliced-ELL product for large block-sparse Q (xm_sell.h) -------------------------------------------------------------
int xm_sell_layout(const int64_t *rowptr, const int32_t *colidx, int64_t n, int64_t ncols, int slabs, int lmax, int64_t sizes[5],
                   int64_t *slice_off, int32_t *slab_start, uint8_t *kind, int64_t *src, int32_t *pslot, int64_t *pptr, int32_t *ridx) {
    XM_TRY
    xm::SellHost h;
    xm::sell_build_host(rowptr, colidx, n, ncols, slabs, lmax, h);   // host only: no device needed
    if (sizes) { sizes[0] = h.nslices; sizes[1] = h.nsteps; sizes[2] = h.nparts; sizes[3] = h.nvrows; sizes[4] = h.nstore; }
    if (slice_off) std::copy(h.slice_off.begin(), h.slice_off.end(), slice_off);
    if (slab_start) std::copy(h.slab_start.begin(), h.slab_start.end(), slab_start);
    if (kind) std::copy(h.kind.begin(), h.kind.end(), kind);
    if (src) std::copy(h.src.begin(), h.src.end(), src);
    if (pslot) std::copy(h.pslot.begin(), h.pslot.end(), pslot);
    if (pptr) std::copy(h.pptr.begin(), h.pptr.end(), pptr);
    if (ridx) std::copy(h.ridx.begin(), h.ridx.begin() + h.nparts, ridx);
    return XM_OK;
    XM_CATCH
}
// host-only: the column-locality figures behind the automatic choice of xm_tuning_t.sell_wpad (SellHost::lines_*):
// lines[0..2] = distinct 128-byte lines of W per sampled step at the native pitch of 72-byte records, of 120-byte records, at the 128-byte pitch
int xm_sell_locality(const int64_t *rowptr, const int32_t *colidx, int64_t n, int64_t ncols, int slabs, int lmax, int64_t lines[3]) {
    XM_TRY
    if (!lines) throw xm::Error(XM_ERR_ARG, "null output");
    xm::SellHost h;
    xm::sell_build_host(rowptr, colidx, n, ncols, slabs, lmax, h);
    lines[0] = h.lines_native72; lines[1] = h.lines_native120; lines[2] = h.lines_padded;
    return XM_OK;
    XM_CATCH
}
// which transport joins the ranks of this context: 0 none (one GPU) | 1 RCCL | 2 shared-memory test transport | 3 direct peer writes between the
// host threads of this process | 4 direct peer writes between processes (IPC); note = why a faster transport was given up (empty: it was not)
int xm_ctx_transport(xm_ctx_t *ctx, int *kind, char *note, size_t note_cap) {
    XM_TRY
    if (!ctx || (!ctx->impl && !ctx->team)) throw xm::Error(XM_ERR_ARG, "null context");
    const int k = ctx->team ? ctx->team->comm_kind() : ctx->impl->comm_kind();
    const std::string &n = ctx->team ? ctx->team->fallback_note() : ctx->impl->fallback_note();
    if (kind) *kind = k;
    if (note && note_cap > 0) { std::strncpy(note, n.c_str(), note_cap - 1); note[note_cap - 1] = 0; }
    return XM_OK;
    XM_CATCH
}

int xm_ctx_product_kind(xm_ctx_t *ctx, int o, int *kind) {
    XM_TRY
    if (!ctx || (!ctx->impl && !ctx->team) || !kind) throw xm::Error(XM_ERR_ARG, "null argument");
    *kind = ctx->team ? ctx->team->product_kind(o) : ctx->impl->product_kind(o);
    return XM_OK;
    XM_CATCH
}

// 1 when the tCG of the last solved rank kept its product input at the 128-byte record pitch as well (xm_tuning_t.sell_wpad); 0 otherwise / several GPUs
int xm_ctx_sell_wpad(xm_ctx_t *ctx, int *on) {
    XM_TRY
    if (!ctx || (!ctx->impl && !ctx->team) || !on) throw xm::Error(XM_ERR_ARG, "null argument");
    *on = (ctx->impl && ctx->impl->sell_wpad_on()) ? 1 : 0;
    return XM_OK;
    XM_CATCH
}

int xm_sell_create2(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t n, int64_t ncols, int slabs, int lmax,
                    int codec, int64_t row0, void **handle) {
    XM_TRY
    require_device();
    if (!rowptr || !handle || n < 1 || row0 < 0) throw xm::Error(XM_ERR_ARG, "bad argument");
    *handle = new xm::SellMatrix(rowptr, colidx, blocks, n, ncols, slabs, lmax > 0 ? lmax : 64, nullptr, codec, row0);
    return XM_OK;
    XM_CATCH
}
int xm_sell_create(const int64_t *rowptr, const int32_t *colidx, const double *blocks, int64_t n, int64_t ncols, int slabs, int lmax,
                   void **handle) {
    return xm_sell_create2(rowptr, colidx, blocks, n, ncols, slabs, lmax, 0, 0, handle);
}
int xm_sell_quat_roundtrip(const double block[9], double quat[4], double rebuilt[9]) {
    XM_TRY
    if (!block || !quat || !rebuilt) throw xm::Error(XM_ERR_ARG, "null argument");
    xm::sell_quat_roundtrip(block, quat, rebuilt);
    return XM_OK;
    XM_CATCH
}
void xm_sell_destroy(void *handle) { delete static_cast<xm::SellMatrix *>(handle); }
int xm_qw_sell_padded(void *handle, int o, const double *dW, const double *dWpad16, double *dOut, double alpha, int gather_mode, void *stream) {
    XM_TRY
    if (!handle) throw xm::Error(XM_ERR_ARG, "null handle");
    if (gather_mode != 0 && gather_mode != 1) throw xm::Error(XM_ERR_ARG, "gather_mode must be 0 or 1");
    xm::SellMatrix &m = *static_cast<xm::SellMatrix *>(handle);
    xm::launch_qw_sell(o, xm::EPI_PLAIN, m, dW, alpha, plain_args(m.nloc(), dOut), gather_mode, (hipStream_t)stream, dWpad16);
    return XM_OK;
    XM_CATCH
}
int xm_qw_sell(void *handle, int o, const double *dW, double *dOut, double alpha, int gather_mode, void *stream) {
    return xm_qw_sell_padded(handle, o, dW, nullptr, dOut, alpha, gather_mode, stream);
}

// symmetric r x r eigen-decomposition (cyclic Jacobi), ascending; V columns = eigenvectors (col-major)
static void jacobi_eig(int r, std::vector<double> &A, std::vector<double> &V, std::vector<double> &w) {
    V.assign((size_t)r * r, 0.0);
    for (int i = 0; i < r; ++i) V[i + (size_t)i * r] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < r; ++p) for (int q = p + 1; q < r; ++q) off += A[p + (size_t)q * r] * A[p + (size_t)q * r];
        if (off < 1e-300) break;
        for (int p = 0; p < r; ++p)
            for (int q = p + 1; q < r; ++q) {
                const double apq = A[p + (size_t)q * r];
                if (apq == 0.0) continue;
                const double theta = (A[q + (size_t)q * r] - A[p + (size_t)p * r]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < r; ++k) {
                    const double akp = A[k + (size_t)p * r], akq = A[k + (size_t)q * r];
                    A[k + (size_t)p * r] = c * akp - sn * akq; A[k + (size_t)q * r] = sn * akp + c * akq;
                }
                for (int k = 0; k < r; ++k) {
                    const double apk = A[p + (size_t)k * r], aqk = A[q + (size_t)k * r];
                    A[p + (size_t)k * r] = c * apk - sn * aqk; A[q + (size_t)k * r] = sn * apk + c * aqk;
                }
                for (int k = 0; k < r; ++k) {
                    const double vkp = V[k + (size_t)p * r], vkq = V[k + (size_t)q * r];
                    V[k + (size_t)p * r] = c * vkp - sn * vkq; V[k + (size_t)q * r] = sn * vkp + c * vkq;
                }
            }
    }
    w.resize((size_t)r);
    for (int i = 0; i < r; ++i) w[(size_t)i] = A[i + (size_t)i * r];
}

}  // extern "C"
// variant: kernel form of the per-camera projection (launch_recover_project); reps > 0 and ms_avg: that launch timed with HIP events
void xm::recover_rotations(int64_t n, int r, const double *R, const double *s, double *rot, double *scale, int *n_negative_det, int variant,
                           int reps, double *ms_avg) {
    require_device();
    if (n < 1 || r < 3 || r > 16 || !R || !s || !rot || !scale || variant < 0 || variant > 1) throw xm::Error(XM_ERR_ARG, "bad argument");
    const int64_t m = 3 * n;
    xm::DevBuf<double> dR, ds, dV, drot, dscale, dparts;
    xm::DevBuf<int> dneg;
    dR.alloc((size_t)m * r, false); ds.alloc((size_t)n, false); dV.alloc((size_t)r * 3); drot.alloc((size_t)9 * n, false);
    dscale.alloc((size_t)n, false); dneg.alloc(1);
    XM_HIP_CHECK(hipMemcpy(dR.p, R, (size_t)m * r * sizeof(double), hipMemcpyHostToDevice));
    XM_HIP_CHECK(hipMemcpy(ds.p, s, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> V((size_t)r * 3, 0.0);
    if (r == 3) {
        V[0] = V[4] = V[8] = 1.0;
    } else {
        // top-3 eigenvectors of the r x r Gram matrix sR^T sR == top-3 right singular vectors of sR; sR*V spans the same
        // rank-3 factor as the reference's eigh of the 3n x 3n matrix sR sR^T (recoversolution.py:12-24) up to a 3x3 orthogonal
        // gauge, which the anchoring removes.
        const int grid = xm::flat_grid(m);
        dparts.alloc((size_t)grid * r * r);
        xm::launch_recover_gram(n, r, dR.p, ds.p, dparts.p, grid, nullptr);
        std::vector<double> parts((size_t)grid * r * r), G((size_t)r * r, 0.0), Ev, w;
        XM_HIP_CHECK(hipMemcpy(parts.data(), dparts.p, parts.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int b = 0; b < grid; ++b) for (int e = 0; e < r * r; ++e) G[(size_t)e] += parts[(size_t)b * r * r + e];
        jacobi_eig(r, G, Ev, w);
        std::vector<int> idx((size_t)r);
        for (int i = 0; i < r; ++i) idx[(size_t)i] = i;
        std::sort(idx.begin(), idx.end(), [&](int a, int b) { return w[(size_t)a] > w[(size_t)b]; });
        for (int c = 0; c < 3; ++c) for (int k = 0; k < r; ++k) V[k + (size_t)c * r] = Ev[k + (size_t)idx[(size_t)c] * r];
    }
    XM_HIP_CHECK(hipMemcpy(dV.p, V.data(), V.size() * sizeof(double), hipMemcpyHostToDevice));
    xm::launch_recover_project(n, r, dR.p, ds.p, dV.p, drot.p, dscale.p, dneg.p, nullptr, variant);
    if (reps > 0 && ms_avg) {
        hipEvent_t e0, e1;
        XM_HIP_CHECK(hipEventCreate(&e0)); XM_HIP_CHECK(hipEventCreate(&e1));
        XM_HIP_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) {
            XM_HIP_CHECK(hipMemsetAsync(dneg.p, 0, sizeof(int), nullptr));
            xm::launch_recover_project(n, r, dR.p, ds.p, dV.p, drot.p, dscale.p, dneg.p, nullptr, variant);
        }
        XM_HIP_CHECK(hipEventRecord(e1, nullptr));
        XM_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        XM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms_avg = (double)ms / reps;
    }
    int neg = 0;
    XM_HIP_CHECK(hipMemcpy(&neg, dneg.p, sizeof(int), hipMemcpyDeviceToHost));
    if (2 * (int64_t)neg > n) xm::launch_negate(drot.p, 9 * n, nullptr);   // recoversolution.py:60-62 (polar(-M) = -polar(M))
    XM_HIP_CHECK(hipMemcpy(rot, drot.p, (size_t)9 * n * sizeof(double), hipMemcpyDeviceToHost));
    XM_HIP_CHECK(hipMemcpy(scale, dscale.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (n_negative_det) *n_negative_det = neg;
}
extern "C" {
int xm_recover_rotations(int64_t n, int r, const double *R, const double *s, double *rot, double *scale, int *n_negative_det) {
    XM_TRY
    xm::recover_rotations(n, r, R, s, rot, scale, n_negative_det, 0, 0, nullptr);
    return XM_OK;
    XM_CATCH
}

int xm_ctx_edge_residuals_recovered(xm_ctx_t *ctx, const double *rot, const double *scale, double *res) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    if (ctx->team) ctx->team->edge_residuals_recovered(rot, scale, res); else ctx->impl->edge_residuals_recovered(rot, scale, res);
    return XM_OK;
    XM_CATCH
}
int xm_ctx_xm2_filter(xm_ctx_t *ctx, const double *rot, const double *scale, double percentile, double *threshold, int64_t *removed, double *w_out) {
    XM_TRY
    if (!ctx) throw xm::Error(XM_ERR_ARG, "null argument");
    const double thr = ctx->team ? ctx->team->xm2_filter(rot, scale, percentile, removed, w_out) : ctx->impl->xm2_filter(rot, scale, percentile, removed, w_out);
    if (threshold) *threshold = thr;
    return XM_OK;
    XM_CATCH
}
int xm_ctx_xm2_round(xm_ctx_t *ctx, const double *R, const double *s, int r, const xm_options_t *opt, xm_xm2_info_t *info, xm_result_t *res) {
    XM_TRY
    if (!ctx || !R || !s || !opt || !info || !res) throw xm::Error(XM_ERR_ARG, "null argument");
    // a multi-GPU context fans every step out to its ranks (each evaluates the filter itself: identical numbers, no exchange)
    auto solve_ctx = [&](const xm_options_t &o_, xm_result_t &r_) { if (ctx->team) ctx->team->solve(o_, r_); else ctx->impl->solve(o_, r_); };
    xm_options_t op = take_struct(opt, "xm_options_t");
    xm_result_t rs = take_struct(res, "xm_result_t");
    xm_xm2_info_t inf = take_struct(info, "xm_xm2_info_t");
    const uint32_t caller_res = res->struct_size, caller_inf = info->struct_size;
    const int64_t n = ctx->team ? ctx->team->cameras() : ctx->impl->cameras();
    // recover_XM's rotations and scales of the starting solution (utils/recoversolution.py:12-86)
    std::vector<double> rot((size_t)9 * n), scale((size_t)n);
    int neg = 0;
    if (xm_recover_rotations(n, r, R, s, rot.data(), scale.data(), &neg) != XM_OK) throw xm::Error(XM_ERR_HIP, g_err);
    const double pct = (inf.percentile > 0.0) ? inf.percentile : 90.0;
    inf.threshold = ctx->team ? ctx->team->xm2_filter(rot.data(), scale.data(), pct, &inf.removed, nullptr)
                              : ctx->impl->xm2_filter(rot.data(), scale.data(), pct, &inf.removed, nullptr);
    int64_t kept = 0;
    for (double w : (ctx->team ? ctx->team->weights() : ctx->impl->weights())) kept += (w != 0.0);
    // second pass: rank-3 solve without regulariser, then decide on lam from the spread of its scales (3_test_colmap_glomap.py:339-351)
    std::vector<double> R3((size_t)3 * n * 4, 0.0), s3((size_t)n, 1.0);
    xm_options_t o3 = op;
    o3.mode = XM_MODE_RANK3; o3.lam = 0.0; o3.max_rank = 3; o3.flags &= ~XM_FLAG_WARM_R; o3.R_ini = nullptr; o3.s_ini = nullptr; o3.trace = nullptr; o3.trace_cap = 0;
    xm_result_t r3;
    std::memset(&r3, 0, sizeof(r3));
    r3.R = R3.data(); r3.s = s3.data();
    solve_ctx(o3, r3);
    inf.rank3_status = r3.status; inf.rank3_tcg_iters = r3.tcg_iters;
    double mean = 0.0, var = 0.0;
    int64_t small = 0;
    for (int64_t i = 1; i < n; ++i) mean += s3[(size_t)i];
    mean = (n > 1) ? mean / (double)(n - 1) : 1.0;
    for (int64_t i = 1; i < n; ++i) var += (s3[(size_t)i] - mean) * (s3[(size_t)i] - mean);
    const double sd = (n > 1) ? std::sqrt(var / (double)(n - 1)) : 0.0;   // numpy.std: population standard deviation
    for (int64_t i = 0; i < n; ++i) small += (s3[(size_t)i] < 0.1);
    inf.s_avg = mean; inf.s_std = sd; inf.n_small = small;
    inf.regularised = (std::fabs(mean - 1.0) > 2.0 * sd || small > 10) ? 1 : 0;
    inf.lam_used = inf.regularised ? (double)kept / (double)n : 0.0;
    xm_options_t of = op;
    of.lam = inf.lam_used;
    if (op.flags & XM_FLAG_WARM_R) { of.mode = XM_MODE_REBUTTLE; of.R_ini = R3.data(); of.s_ini = s3.data(); }
    else { of.mode = XM_MODE_SOLVE; of.R_ini = nullptr; of.s_ini = nullptr; }
    solve_ctx(of, rs);
    give_result(res, rs, caller_res);
    inf.struct_size = caller_inf;
    std::memcpy(info, &inf, std::min<size_t>(caller_inf, sizeof(inf)));
    return XM_OK;
    XM_CATCH
}

int xm_comm_unique_id(unsigned char id[128]) { XM_TRY xm::comm_unique_id(id); return XM_OK; XM_CATCH }
int xm_comm_init(int rank, int world, int device, const unsigned char id[128], const char *rccl_path) {
    XM_TRY require_device(); xm::comm_init(rank, world, device, id, rccl_path); return XM_OK; XM_CATCH
}
int xm_comm_init_shm(int rank, int world, int device, const char *name, size_t bytes) {
    XM_TRY require_device(); xm::comm_init_shm(rank, world, device, name, bytes); return XM_OK; XM_CATCH
}
int xm_comm_init_ipc(int rank, int world, int device, const char *name, double spin_seconds) {
    XM_TRY require_device(); xm::comm_init_ipc(rank, world, device, name, spin_seconds); return XM_OK; XM_CATCH
}
int xm_comm_finalize(void) { XM_TRY xm::comm_finalize(); return XM_OK; XM_CATCH }
int xm_partition(int64_t n, int world, int rank, int64_t *c0, int64_t *c1) {
    if (n < 0 || world < 1 || rank < 0 || rank >= world || !c0 || !c1) { g_err = "bad argument"; return XM_ERR_ARG; }
    const int64_t per = xm::equal_range_len(n, world);
    *c0 = std::min<int64_t>(n, (int64_t)rank * per);
    *c1 = std::min<int64_t>(n, (int64_t)(rank + 1) * per);
    return XM_OK;
}

int xm_partition_blocks(int64_t n, const int64_t *rowptr, int world, int rank, int64_t *c0, int64_t *c1) {
    XM_TRY
    if (n < 0 || !rowptr || world < 1 || rank < 0 || rank >= world || !c0 || !c1) throw xm::Error(XM_ERR_ARG, "bad argument");
    std::vector<int64_t> cuts;
    xm::partition_cuts(n, world, rowptr, cuts);
    *c0 = cuts[(size_t)rank]; *c1 = cuts[(size_t)rank + 1];
    return XM_OK;
    XM_CATCH
}

}  // extern "C"

// host-only view of the multi-rank symmetric window plan (xm_symw.h) for the CPU test: geom = {T, Th, tie, t0, nsteps, nstrips, K, items};
// items: 3 ints each (strip, jb, je), NULL = only the sizes
int xm_symw_plan(int64_t ntot, int nloc, int cam0, int K, int32_t geom[8], int32_t *items) {
    XM_TRY
    xm::SymwPlan p;
    xm::symw_plan_build(ntot, nloc, cam0, K, p);
    if (geom) {
        geom[0] = p.g.T; geom[1] = p.g.Th; geom[2] = p.g.tie; geom[3] = p.g.t0; geom[4] = p.g.nsteps; geom[5] = p.g.nstrips; geom[6] = p.K;
        geom[7] = (int32_t)p.items.size();
    }
    if (items)
        for (size_t i = 0; i < p.items.size(); ++i) { items[3 * i] = p.items[i].s; items[3 * i + 1] = p.items[i].jb; items[3 * i + 2] = p.items[i].je; }
    return XM_OK;
    XM_CATCH
}
// block (t, u) used by row step t? (the predicate of the sweep's masks)
int xm_symw_use(int T, int t, int u) {
    xm::SymwGeom g;
    g.T = T; g.Th = (T + 1) / 2; g.tie = (T % 2 == 0) ? 1 : 0; g.t0 = 0; g.nsteps = T; g.nstrips = (6 * T + xm::kSwStrip - 1) / xm::kSwStrip;
    return xm::symw_use(g, t, u) ? 1 : 0;
}

