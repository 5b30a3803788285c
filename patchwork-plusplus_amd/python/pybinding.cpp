// pybinding.cpp -- Python module `pypatchworkpp` with the reference's API surface
// (/root/reference/python/patchworkpp/pybinding.cpp:9-57: classes `Parameters` and
// `patchworkpp`, same attribute and method names), backed by the MI355X library.
//
// The reference binds Eigen types through pybind11/eigen.h; there is no Eigen here, so
// arrays cross as numpy buffers directly: estimateGround() takes any 2-D array convertible
// to float32 (C- or F-contiguous is used in place, anything else is copied), the getters
// return fresh C-contiguous numpy arrays: (n, 3) float32 and (n,) int32.  The GIL is
// released while the GPU works.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "patchwork/patchworkpp.h"

namespace py = pybind11;
using patchwork::Params;
using patchwork::PatchWorkpp;

namespace {

py::array_t<float> to_numpy(const patchwork::Cloud &c) {
    py::array_t<float> out({(py::ssize_t)c.rows(), (py::ssize_t)3});
    if (c.rows() > 0) std::memcpy(out.mutable_data(), c.data(), (size_t)c.rows() * 3 * sizeof(float));
    return out;
}
py::array_t<int32_t> to_numpy(const patchwork::Indices &v) {
    py::array_t<int32_t> out((py::ssize_t)v.rows());
    if (v.rows() > 0) std::memcpy(out.mutable_data(), v.data(), (size_t)v.rows() * sizeof(int32_t));
    return out;
}

void estimate_ground(PatchWorkpp &self, py::array cloud) {
    if (cloud.ndim() != 2) throw py::value_error("estimateGround expects a 2-D array (N, 3|4)");
    // F-contiguous float32 is consumed as column-major (the layout Eigen::MatrixXf would have),
    // everything else is brought to C-contiguous float32 (no copy when it already is)
    const bool f_order = (cloud.flags() & py::array::f_style) && !(cloud.flags() & py::array::c_style) &&
                         cloud.dtype().is(py::dtype::of<float>());
    py::array a;
    if (f_order)
        a = py::array_t<float, py::array::f_style>::ensure(cloud);
    else
        a = py::array_t<float, py::array::c_style | py::array::forcecast>::ensure(cloud);
    if (!a) throw py::value_error("estimateGround: cannot convert the input to float32");
    const float *data = static_cast<const float *>(a.data());
    const int rows = (int)a.shape(0), cols = (int)a.shape(1);
    py::gil_scoped_release release;
    self.estimateGround(data, rows, cols, !f_order);
}

}  // namespace

#define PWPP_FIELD(name) cls.def_readwrite(#name, &Params::name)

PYBIND11_MODULE(pypatchworkpp, m) {
    m.doc() = "Python Patchwork++ (MI355X / HIP backend)";
    m.attr("__version__") = "0.0.1";
    m.attr("backend") = "hip-gfx950";

    {
        py::class_<Params> cls(m, "Parameters");
        cls.def(py::init<>());
        PWPP_FIELD(verbose);
        PWPP_FIELD(enable_RNR);
        PWPP_FIELD(enable_RVPF);
        PWPP_FIELD(enable_TGR);
        PWPP_FIELD(num_iter);
        PWPP_FIELD(num_lpr);
        PWPP_FIELD(num_min_pts);
        PWPP_FIELD(num_zones);
        PWPP_FIELD(num_rings_of_interest);
        PWPP_FIELD(RNR_ver_angle_thr);
        PWPP_FIELD(RNR_intensity_thr);
        PWPP_FIELD(sensor_height);
        PWPP_FIELD(th_seeds);
        PWPP_FIELD(th_dist);
        PWPP_FIELD(th_seeds_v);
        PWPP_FIELD(th_dist_v);
        PWPP_FIELD(max_range);
        PWPP_FIELD(min_range);
        PWPP_FIELD(uprightness_thr);
        PWPP_FIELD(adaptive_seed_selection_margin);
        PWPP_FIELD(intensity_thr);
        PWPP_FIELD(num_sectors_each_zone);
        PWPP_FIELD(num_rings_each_zone);
        PWPP_FIELD(max_flatness_storage);
        PWPP_FIELD(max_elevation_storage);
        PWPP_FIELD(elevation_thr);
        PWPP_FIELD(flatness_thr);
    }

    py::class_<PatchWorkpp>(m, "patchworkpp")
        .def(py::init<Params>())
        .def(py::init<Params, int>(), py::arg("params"), py::arg("device"))
        .def("estimateGround", &estimate_ground)
        .def("getHeight", &PatchWorkpp::getHeight)
        .def("getTimeTaken", &PatchWorkpp::getTimeTaken)
        .def("setReferenceOrder", &PatchWorkpp::setReferenceOrder, py::arg("on"))
        .def("getGround", [](PatchWorkpp &s) { return to_numpy(s.getGround()); })
        .def("getNonground", [](PatchWorkpp &s) { return to_numpy(s.getNonground()); })
        .def("getCenters", [](PatchWorkpp &s) { return to_numpy(s.getCenters()); })
        .def("getNormals", [](PatchWorkpp &s) { return to_numpy(s.getNormals()); })
        .def("getGroundIndices", [](PatchWorkpp &s) { return to_numpy(s.getGroundIndices()); })
        .def("getNongroundIndices", [](PatchWorkpp &s) { return to_numpy(s.getNongroundIndices()); });
}
