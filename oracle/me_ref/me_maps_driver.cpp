// TEST INFRASTRUCTURE - a driver around the REAL MinkowskiEngine coordinate-map manager (CPU), compiled by oracle/build_me_ref.py
// together with /root/reference/MinkowskiEngine/src/coordinate_map_manager.cpp (where it lies; -DCPU_ONLY, no BLAS involved) into
// oracle/_ref/me_maps.so.  This file contains no MinkowskiEngine code: it calls the reference's CoordinateMapManager exactly as
// its convolution front ends do and hands the resulting coordinate maps and kernel maps back as plain tensors, so that
// oracle/fcgf_oracle.py's restatement of those semantics can be pinned against the reference itself (oracle/gen_golden_me.py
// -> tests/golden/me_maps.npz).  The calls mirrored:
//   input map        SparseTensor construction            MinkowskiEngine/MinkowskiSparseTensor.py -> manager.insert_and_map
//   strided conv     ConvolutionForwardCPU                src/convolution_cpu.cpp:76-118   (out key = manager.stride(in, stride))
//   transposed conv  ConvolutionTransposeForwardCPU       src/convolution_transpose_cpu.cpp:75-107 (out key = stride_region on the
//                                                          divided tensor stride, no new coordinates; kernel map is_transpose = true)
// What is NOT built: the convolution arithmetic itself (src/math_functions_cpu.cpp needs cblas.h, absent from the image); given the
// maps it is out[o] += in[i] W[k] over the pairs of kernel index k, which torch's dense convolutions pin (tests/test_fcgf_oracle.py).
#include "coordinate_map_cpu.hpp"
#include "coordinate_map_key.hpp"
#include "coordinate_map_manager.hpp"
#include "kernel_region.hpp"
#include "types.hpp"
#include "utils.hpp"

#include <torch/extension.h>

#include <string>
#include <vector>

namespace mk = minkowski;
using manager_t = mk::cpu_manager_type<int32_t>;
using stride_t = mk::default_types::stride_type;

static py::list map_to_list(mk::cpu_kernel_map const& km) {
    // one (in rows, out rows) pair of int32 tensors per kernel index, in kernel-index order
    py::list out;
    for (size_t k = 0; k < km.first.size(); ++k) {
        auto const n = (long)km.first[k].size();
        at::Tensor in = torch::empty({n}, torch::TensorOptions().dtype(torch::kInt32));
        at::Tensor ou = torch::empty({n}, torch::TensorOptions().dtype(torch::kInt32));
        std::copy_n(km.first[k].data(), n, in.data_ptr<int32_t>());
        std::copy_n(km.second[k].data(), n, ou.data_ptr<int32_t>());
        out.append(py::make_tuple(in, ou));
    }
    return out;
}

// coords: (N, 4) int32 rows (batch, x, y, z), duplicates allowed.  Returns a dict with the unique / inverse maps of the insertion,
// the coordinates of the tensor-stride 1, 2, 4, 8 maps (rows in the manager's order) and the kernel maps of every convolution of
// fcgf_model/resunet.py's ResUNetBN2C: conv1 (k = conv1_kernel, stride 1), the 3^3 stride-1 maps of the four levels (block*),
// conv2 / conv3 / conv4 (k = 3, stride 2) and conv4_tr / conv3_tr / conv2_tr (transposed, k = 3, stride 2).
static py::dict fcgf_maps(at::Tensor coords, int conv1_kernel) {
    TORCH_CHECK(coords.dim() == 2 && coords.size(1) == 4 && coords.scalar_type() == torch::kInt32 && coords.is_contiguous(),
                "coords must be a contiguous (N, 4) int32 tensor");
    manager_t mgr;
    py::dict out;
    auto ins = mgr.insert_and_map(coords, stride_t{1, 1, 1}, "");
    mk::CoordinateMapKey* key1 = py::cast<mk::CoordinateMapKey*>(ins.first);
    out["unique_map"] = ins.second.first;
    out["inverse_map"] = ins.second.second;

    stride_t const k3{3, 3, 3}, s1{1, 1, 1}, s2{2, 2, 2}, d1{1, 1, 1};
    at::Tensor const offset = torch::empty({0}, torch::TensorOptions().dtype(torch::kInt32));
    std::vector<mk::CoordinateMapKey> keys;
    keys.reserve(4);
    keys.push_back(*key1);
    // strided convolutions: conv2 (1 -> 2), conv3 (2 -> 4), conv4 (4 -> 8)
    for (int l = 0; l < 3; ++l) {
        auto out_key = std::get<0>(mgr.stride(keys[l].get_key(), s2));
        mk::CoordinateMapKey ok(4);
        ok.set_key(out_key);
        keys.push_back(ok);
        out[py::str("conv_s2_" + std::to_string(l))] = map_to_list(
            mgr.kernel_map(&keys[l], &keys[l + 1], k3, s2, d1, mk::RegionType::HYPER_CUBE, offset, false, false));
    }
    for (int l = 0; l < 4; ++l) {
        out[py::str("coords_" + std::to_string(l))] = mgr.get_coordinates(&keys[l]);
        out[py::str("conv_s1_" + std::to_string(l))] =
            map_to_list(mgr.kernel_map(&keys[l], &keys[l], k3, s1, d1, mk::RegionType::HYPER_CUBE, offset, false, false));
    }
    {
        stride_t const kk{(unsigned)conv1_kernel, (unsigned)conv1_kernel, (unsigned)conv1_kernel};
        out["conv1"] = map_to_list(mgr.kernel_map(&keys[0], &keys[0], kk, s1, d1, mk::RegionType::HYPER_CUBE, offset, false, false));
    }
    // transposed convolutions: conv4_tr (8 -> 4), conv3_tr (4 -> 2), conv2_tr (2 -> 1); the out key as convolution_transpose_cpu.cpp
    // derives it when the caller gives none (generate_new_coordinates = false)
    for (int l = 3; l >= 1; --l) {
        auto it = mgr.find(keys[l].get_key());
        TORCH_CHECK(it != mgr.map_end(), "map not found");
        auto const& in_map = (*it).second;
        auto out_ts = mk::detail::stride_tensor_stride(in_map.get_tensor_stride(), s2, true);
        auto region = mk::cpu_kernel_region<int32_t>(mk::RegionType::HYPER_CUBE, in_map.coordinate_size(), out_ts.data(), k3.data(), d1.data(), 0,
                                                     offset.data_ptr<int32_t>(), offset.size(0), true);
        auto out_key = std::get<0>(mgr.stride_region(keys[l].get_key(), region, out_ts, false));
        mk::CoordinateMapKey ok(4);
        ok.set_key(out_key);
        out[py::str("tr_out_is_level_" + std::to_string(l))] = (out_key == keys[l - 1].get_key());
        out[py::str("tr_out_coords_" + std::to_string(l))] = mgr.get_coordinates(&ok);
        out[py::str("conv_tr_" + std::to_string(l))] =
            map_to_list(mgr.kernel_map(&keys[l], &ok, k3, s2, d1, mk::RegionType::HYPER_CUBE, offset, true, false));
    }
    return out;
}

// the 2-D figure of tests/python/kernel_map.py (any dimension): kernel map of a kernel-3, stride-2 convolution on coords (N, D + 1)
static py::dict strided_map_nd(at::Tensor coords) {
    TORCH_CHECK(coords.dim() == 2 && coords.scalar_type() == torch::kInt32 && coords.is_contiguous(), "coords: contiguous (N, D + 1) int32");
    unsigned const D = (unsigned)coords.size(1) - 1;
    manager_t mgr;
    auto ins = mgr.insert_and_map(coords, stride_t(D, 1), "");
    mk::CoordinateMapKey* key1 = py::cast<mk::CoordinateMapKey*>(ins.first);
    auto out_key = std::get<0>(mgr.stride(key1->get_key(), stride_t(D, 2)));
    mk::CoordinateMapKey ok(D + 1);
    ok.set_key(out_key);
    at::Tensor const offset = torch::empty({0}, torch::TensorOptions().dtype(torch::kInt32));
    py::dict out;
    out["in_coords"] = mgr.get_coordinates(key1);
    out["out_coords"] = mgr.get_coordinates(&ok);
    out["map"] = map_to_list(mgr.kernel_map(key1, &ok, stride_t(D, 3), stride_t(D, 2), stride_t(D, 1), mk::RegionType::HYPER_CUBE, offset, false, false));
    return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    py::class_<mk::CoordinateMapKey>(m, "CoordinateMapKey").def("__repr__", &mk::CoordinateMapKey::to_string);
    m.def("fcgf_maps", &fcgf_maps, "coordinate maps and kernel maps of the FCGF backbone from MinkowskiEngine's CPU coordinate manager");
    m.def("strided_map_nd", &strided_map_nd, "kernel-3 stride-2 kernel map in any dimension");
}
