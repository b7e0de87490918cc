// xm_pybind.cpp — Python module `XM`: the reference's pybind11 surface (XM/src/XM_main.cu:403-408) over the C ABI.
//   XM.solve(dataset_path, max_rank, tol, lam, max_time)          -> None
//   XM.solve_rebuttle(dataset_path, max_rank, tol, lam, max_time) -> int
//   XM.solve_rank3(dataset_path, max_rank, tol, lam, max_time)    -> None
// Like the reference, no py::arg names/defaults are registered: all five arguments are positional and required.
// Unlike the reference (which prints CUDA errors and carries on, Utils/check.h:41-76), I/O and HIP failures raise
// RuntimeError; numerical non-convergence does not raise (R.bin / s.bin are still written).  The GIL is released
// while the GPU solve runs.
#include <pybind11/pybind11.h>

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

PYBIND11_MODULE(XM, m) {
    m.doc() = "pybind11 for XM (MI355X-native build)";
    m.def("solve", &solve, "XM main function");
    m.def("solve_rebuttle", &solve_rebuttle, "permit give initial guess");
    m.def("solve_rank3", &solve_rank3, "XM main function for rank 3 only");
}
