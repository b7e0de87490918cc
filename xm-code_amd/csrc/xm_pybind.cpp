// xm_pybind.cpp — Python module `XM`: the reference's pybind11 surface (XM/src/XM_main.cu:403-408) over the C ABI.
//   XM.solve(dataset_path, max_rank, tol, lam, max_time)          -> None
//   XM.solve_rebuttle(dataset_path, max_rank, tol, lam, max_time) -> int
//   XM.solve_rank3(dataset_path, max_rank, tol, lam, max_time)    -> None
// Additive (not in the reference): the same solve on arrays, without the Q.bin / R.bin round trip
//   XM.solve_array(Q, max_rank, tol, lam, max_time, mode=0, s_ini=None, flags=0, R_ini=None)      -> (R, s, info)
//   XM.solve_bsr(rowptr, colidx, blocks, max_rank, tol, lam, max_time, mode=0, s_ini=None, flags=0) -> (R, s, info)
// both take n_gpus=1, gpu_map=0, retraction=0: the single-process row partition over several GPUs (include/xm_amd.h) and the polar
// retraction; the file functions take the GPU count from the environment (XM_GPUS=8 python 1_test_solve.py).
// Like the reference, no py::arg names/defaults are registered: all five arguments are positional and required.
// Unlike the reference (which prints CUDA errors and carries on, Utils/check.h:41-76), I/O and HIP failures raise
// RuntimeError; numerical non-convergence does not raise (R.bin / s.bin are still written).  The GIL is released
// while the GPU solve runs.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstring>
#include <vector>

#include <stdexcept>
#include <string>

#include "../../include/xm_amd.h"

namespace py = pybind11;

static void raise_on(int rc) {
    if (rc != XM_OK) throw std::runtime_error(std::string("XM: ") + xm_last_error());
}
static void solve(const std::string &dataset_path, unsigned int max_rank, double tol, double lam, double max_time) {
    int rc;
    { py::gil_scoped_release nogil; rc = xm_solve(dataset_path.c_str(), max_rank, tol, lam, max_time); }
    raise_on(rc);
}
static int solve_rebuttle(const std::string &dataset_path, unsigned int max_rank, double tol, double lam, double max_time) {
    int rc, status = 0;
    { py::gil_scoped_release nogil; rc = xm_solve_rebuttle(dataset_path.c_str(), max_rank, tol, lam, max_time, &status); }
    raise_on(rc);
    return status;
}
static void solve_rank3(const std::string &dataset_path, unsigned int max_rank, double tol, double lam, double max_time) {
    int rc;
    { py::gil_scoped_release nogil; rc = xm_solve_rank3(dataset_path.c_str(), max_rank, tol, lam, max_time); }
    raise_on(rc);
}

using darr = py::array_t<double, py::array::f_style | py::array::forcecast>;

static py::tuple solve_problem(xm_problem_t &prob, unsigned int max_rank, double tol, double lam, double max_time, int mode,
                               py::object s_ini, unsigned int flags, py::object R_ini, int n_gpus, int gpu_map, int retraction) {
    prob.struct_size = sizeof(prob);
    prob.n_gpus = n_gpus; prob.gpu_map = gpu_map;
    const int64_t n = prob.n;
    const unsigned rmax = max_rank < 3 ? 3u : max_rank;
    std::vector<double> R((size_t)3 * n * (rmax + 1), 0.0), s((size_t)n, 1.0), sini;
    if (!s_ini.is_none()) {
        darr a = s_ini.cast<darr>();
        if (a.size() < n) throw std::invalid_argument("s_ini needs n entries");
        sini.assign(a.data(), a.data() + n);
    }
    xm_options_t opt;
    std::memset(&opt, 0, sizeof(opt));
    opt.struct_size = sizeof(opt);
    opt.retraction = retraction;
    opt.max_rank = max_rank; opt.tol = tol; opt.lam = lam; opt.max_time = max_time; opt.mode = mode; opt.flags = flags;
    opt.s_ini = sini.empty() ? nullptr : sini.data();
    std::vector<double> rini;
    if (!R_ini.is_none()) {   // warm start of the rank-3 stage (XM_FLAG_WARM_R): first three columns of a previous R
        darr a = R_ini.cast<darr>();
        if (a.ndim() != 2 || a.shape(0) != 3 * n || a.shape(1) < 3) throw std::invalid_argument("R_ini must be 3n x (>= 3)");
        rini.assign(a.data(), a.data() + (size_t)3 * n * 3);
        opt.R_ini = rini.data(); opt.flags |= XM_FLAG_WARM_R;
    }
    xm_result_t res;
    std::memset(&res, 0, sizeof(res));
    res.struct_size = sizeof(res);
    res.R = R.data(); res.s = s.data();
    int rc;
    {
        py::gil_scoped_release nogil;
        xm_ctx_t *ctx = nullptr;
        rc = xm_ctx_create(&prob, &ctx);
        if (rc == XM_OK) { rc = xm_ctx_solve(ctx, &opt, &res); xm_ctx_destroy(ctx); }
    }
    raise_on(rc);
    darr Rout({(py::ssize_t)(3 * n), (py::ssize_t)res.rank});
    std::memcpy(Rout.mutable_data(), R.data(), (size_t)3 * n * res.rank * sizeof(double));
    py::array_t<double> sout((py::ssize_t)n);
    std::memcpy(sout.mutable_data(), s.data(), (size_t)n * sizeof(double));
    py::dict info;
    info["rank"] = res.rank; info["status"] = res.status; info["primal"] = res.primal; info["dual"] = res.dual;
    info["min_eig"] = res.min_eig; info["gap"] = res.gap; info["tcg_iters"] = res.tcg_iters; info["outer_iters"] = res.outer_iters;
    info["qw_products"] = res.qw_products; info["lanczos_iters"] = res.lanczos_iters; info["seconds"] = res.seconds;
    info["cert_flags"] = res.cert_flags; info["eig_residual"] = res.eig_residual; info["n_gpus"] = res.n_gpus; info["exchange"] = res.exchange;
    return py::make_tuple(Rout, sout, info);
}
static py::tuple solve_array(darr Q, unsigned int max_rank, double tol, double lam, double max_time, int mode, py::object s_ini,
                             unsigned int flags, py::object R_ini, int n_gpus, int gpu_map, int retraction) {
    if (Q.ndim() != 2 || Q.shape(0) != Q.shape(1) || Q.shape(0) % 3 != 0 || Q.shape(0) < 3)
        throw std::invalid_argument("Q must be 3n x 3n");
    xm_problem_t prob;
    std::memset(&prob, 0, sizeof(prob));
    prob.n = Q.shape(0) / 3; prob.storage = XM_STORAGE_DENSE; prob.q = Q.data(); prob.ldq = Q.shape(0);
    return solve_problem(prob, max_rank, tol, lam, max_time, mode, s_ini, flags, R_ini, n_gpus, gpu_map, retraction);
}
static py::tuple solve_bsr(py::array_t<int64_t, py::array::c_style | py::array::forcecast> rowptr,
                           py::array_t<int32_t, py::array::c_style | py::array::forcecast> colidx,
                           py::array_t<double, py::array::c_style | py::array::forcecast> blocks, unsigned int max_rank, double tol,
                           double lam, double max_time, int mode, py::object s_ini, unsigned int flags, py::object R_ini, int n_gpus,
                           int gpu_map, int retraction) {
    if (rowptr.size() < 2 || blocks.size() != colidx.size() * 9) throw std::invalid_argument("need rowptr (n+1), colidx (nb), blocks (nb x 3 x 3)");
    xm_problem_t prob;
    std::memset(&prob, 0, sizeof(prob));
    prob.n = rowptr.size() - 1; prob.storage = XM_STORAGE_BSR3; prob.nb = colidx.size();
    prob.rowptr = rowptr.data(); prob.colidx = colidx.data(); prob.blocks = blocks.data();
    if (rowptr.data()[prob.n] != prob.nb) throw std::invalid_argument("rowptr[n] != number of blocks");
    return solve_problem(prob, max_rank, tol, lam, max_time, mode, s_ini, flags, R_ini, n_gpus, gpu_map, retraction);
}

PYBIND11_MODULE(XM, m) {
    m.doc() = "pybind11 for XM (MI355X-native build)";
    m.def("solve", &solve, "XM main function");
    m.def("solve_rebuttle", &solve_rebuttle, "permit give initial guess");
    m.def("solve_rank3", &solve_rank3, "XM main function for rank 3 only");
    m.def("solve_array", &solve_array, "in-memory solve of a dense symmetric Q -> (R, s, info)", py::arg("Q"), py::arg("max_rank"),
          py::arg("tol"), py::arg("lam"), py::arg("max_time"), py::arg("mode") = 0, py::arg("s_ini") = py::none(), py::arg("flags") = 0u,
          py::arg("R_ini") = py::none(), py::arg("n_gpus") = 1, py::arg("gpu_map") = 0, py::arg("retraction") = 0);
    m.def("solve_bsr", &solve_bsr, "in-memory solve of a 3x3-block CSR Q -> (R, s, info)", py::arg("rowptr"), py::arg("colidx"),
          py::arg("blocks"), py::arg("max_rank"), py::arg("tol"), py::arg("lam"), py::arg("max_time"), py::arg("mode") = 0,
          py::arg("s_ini") = py::none(), py::arg("flags") = 0u, py::arg("R_ini") = py::none(), py::arg("n_gpus") = 1, py::arg("gpu_map") = 0,
          py::arg("retraction") = 0);
}
