// segmentation_core.hpp -- everything the Patchwork++ ROS 2 node does between "a sensor_msgs/PointCloud2 arrived" and
// "three PointCloud2 payloads are ready to publish", WITHOUT a single ROS header, so that it can be built and tested where
// there is no ROS (this repository's image).  ros/src/ground_segmentation_server.cpp is the rclcpp component around it.
//
// Behaviour mirrored (reference, read-only under /root/reference):
//   ros/src/GroundSegmentationServer.cpp:24-47   the node's parameters: names, defaults = patchwork::Params, enable_RNR forced off
//   ros/src/GroundSegmentationServer.cpp:76-86   per message: estimateGround, republish the cloud, publish ground / non-ground
//   ros/src/Utils.hpp:158-172                    PointCloud2ToEigenMat: the float32 fields "x", "y", "z" of height * width points
//   ros/src/Utils.hpp:80-104                     CreatePointCloud2Msg: fields x, y, z (FLOAT32 at 0, 4, 8), point_step 16
// Different on purpose: the input message is never converted to a matrix -- the binning kernels read the fields in place
// (pwpp_estimate_ground_fields, include/pwpp.h).
#ifndef PATCHWORKPP_ROS_SEGMENTATION_CORE_HPP
#define PATCHWORKPP_ROS_SEGMENTATION_CORE_HPP

#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "patchwork/patchworkpp.h"

namespace patchworkpp_ros {

constexpr uint8_t kFloat32 = 7;  // sensor_msgs/msg/PointField.FLOAT32

struct Field {  // sensor_msgs/msg/PointField
    std::string name;
    uint32_t offset = 0;
    uint8_t datatype = 0;
    uint32_t count = 1;
};

// what the node reads of an incoming sensor_msgs/msg/PointCloud2 (no copy of the data)
struct CloudView {
    uint32_t height = 1, width = 0, point_step = 0;
    const Field *fields = nullptr;
    size_t num_fields = 0;
    const uint8_t *data = nullptr;
    size_t data_size = 0;
};

// what it fills of an outgoing one: the layout of the reference's CreatePointCloud2Msg (Utils.hpp:80-104)
struct XyzCloud {
    uint32_t height = 1, width = 0, point_step = 16, row_step = 0;
    std::vector<Field> fields;
    std::vector<uint8_t> data;
    static XyzCloud with_points(size_t n) {
        XyzCloud c;
        c.width = (uint32_t)n;
        c.row_step = c.width * c.point_step;
        c.fields = {{"x", 0, kFloat32, 1}, {"y", 4, kFloat32, 1}, {"z", 8, kFloat32, 1}};
        c.data.assign(n * c.point_step, 0);
        return c;
    }
};

// The node's parameters (GroundSegmentationServer.cpp:28-44): `get_double(name, default)` etc. are the node's
// declare_parameter / get_parameter; everything not listed keeps patchwork::Params' default; RNR is off (":46 ToDo. Support intensity").
inline patchwork::Params declare_parameters(const std::function<double(const std::string &, double)> &get_double,
                                            const std::function<int(const std::string &, int)> &get_int,
                                            const std::function<bool(const std::string &, bool)> &get_bool) {
    patchwork::Params p;
    p.sensor_height = get_double("sensor_height", p.sensor_height);
    p.num_iter = get_int("num_iter", p.num_iter);
    p.num_lpr = get_int("num_lpr", p.num_lpr);
    p.num_min_pts = get_int("num_min_pts", p.num_min_pts);
    p.th_seeds = get_double("th_seeds", p.th_seeds);
    p.th_dist = get_double("th_dist", p.th_dist);
    p.th_seeds_v = get_double("th_seeds_v", p.th_seeds_v);
    p.th_dist_v = get_double("th_dist_v", p.th_dist_v);
    p.max_range = get_double("max_range", p.max_range);
    p.min_range = get_double("min_range", p.min_range);
    p.uprightness_thr = get_double("uprightness_thr", p.uprightness_thr);
    p.verbose = get_bool("verbose", p.verbose);
    p.enable_RNR = false;
    return p;
}

class SegmentationCore {
public:
    struct Output {
        XyzCloud cloud, ground, nonground;  // "/patchworkpp/cloud", "/patchworkpp/ground", "/patchworkpp/nonground"
        double time_taken_us = 0.0;
    };

    explicit SegmentationCore(const patchwork::Params &params, int device = 0) : pw_(new patchwork::PatchWorkpp(params, device)) {}

    // one message: GroundSegmentationServer::EstimateGround (:76-86).  The object is long-lived, so the adaptive state carries
    // over from message to message exactly as in the reference node.
    Output estimate(const CloudView &msg) {
        const int ox = offset_of(msg, "x"), oy = offset_of(msg, "y"), oz = offset_of(msg, "z");
        const size_t n = (size_t)msg.height * msg.width;
        if (msg.point_step < 12 || n * msg.point_step > msg.data_size) throw std::runtime_error("PointCloud2: data shorter than height * width * point_step");
        if (n > (size_t)(1 << 22)) throw std::runtime_error("PointCloud2: more than 4194304 points");
        //  every field must lie INSIDE a point: a crafted offset would let the copy below and the device read run past
        // the end of the data
        for (const int o : {ox, oy, oz})
            if (o < 0 || (size_t)o + 4 > (size_t)msg.point_step) throw std::runtime_error("PointCloud2: a field offset lies outside point_step");
        // The library reads the blob in place and wants 4-byte aligned floats (PWPP_E_ARG otherwise).  A message whose step or
        // offsets are not multiples of four -- legal in PointCloud2, unusual for a LiDAR driver -- is repacked to x, y, z first.
        const uint8_t *blob = msg.data;
        int step = (int)msg.point_step, bx = ox, by = oy, bz = oz;
        std::vector<float> repacked;
        if (msg.point_step % 4 != 0 || ox % 4 != 0 || oy % 4 != 0 || oz % 4 != 0 || (reinterpret_cast<uintptr_t>(msg.data) & 3u)) {
            repacked.resize(3 * n);
            for (size_t i = 0; i < n; ++i) {
                const uint8_t *src = msg.data + i * msg.point_step;
                std::memcpy(&repacked[3 * i], src + ox, 4);
                std::memcpy(&repacked[3 * i + 1], src + oy, 4);
                std::memcpy(&repacked[3 * i + 2], src + oz, 4);
            }
            blob = reinterpret_cast<const uint8_t *>(repacked.data());
            step = 12, bx = 0, by = 4, bz = 8;
        }
        const int rc = pwpp_estimate_ground_fields(pw_->handle(), blob, (int)n, step, bx, by, bz, -1);
        if (rc < 0) throw std::runtime_error(std::string("patchworkpp (HIP): ") + pwpp_last_error());
        Output out;
        out.cloud = XyzCloud::with_points(n);  // the reference republishes the cloud as x, y, z (EigenMatToPointCloud2, :80)
        for (size_t i = 0; i < n; ++i) {
            const uint8_t *src = msg.data + i * msg.point_step;
            uint8_t *dst = out.cloud.data.data() + i * 16;
            std::memcpy(dst, src + ox, 4);
            std::memcpy(dst + 4, src + oy, 4);
            std::memcpy(dst + 8, src + oz, 4);
        }
        out.ground = pack(pw_->groundCloud());
        out.nonground = pack(pw_->nongroundCloud());
        out.time_taken_us = pw_->getTimeTaken();
        return out;
    }

    patchwork::PatchWorkpp &patchwork() { return *pw_; }

private:
    std::unique_ptr<patchwork::PatchWorkpp> pw_;

    static int offset_of(const CloudView &msg, const char *name) {
        for (size_t k = 0; k < msg.num_fields; ++k)
            if (msg.fields[k].name == name) {
                if (msg.fields[k].datatype != kFloat32) throw std::runtime_error(std::string("PointCloud2: field ") + name + " is not FLOAT32");
                return (int)msg.fields[k].offset;
            }
        throw std::runtime_error(std::string("PointCloud2: no field ") + name);
    }
    static XyzCloud pack(const patchwork::Cloud &c) {  // FillPointCloud2XYZ (Utils.hpp:118-127)
        XyzCloud out = XyzCloud::with_points((size_t)c.rows());
        for (int i = 0; i < c.rows(); ++i) std::memcpy(out.data.data() + (size_t)i * 16, c.data() + (size_t)i * 3, 12);
        return out;
    }
};

}  // namespace patchworkpp_ros

#endif
