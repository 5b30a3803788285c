// ros_core_demo.cpp -- the ROS 2 node's logic (ros/include/patchworkpp_ros/segmentation_core.hpp) driven without ROS: every
// KITTI .bin file on the command line becomes the payload of a sensor_msgs/PointCloud2 -- 32-byte points with x, y, z at
// offsets 4, 12, 20 between other fields, as a driver might publish them -- and goes through ONE long-lived node core with
// the parameter set of the launch file (ros/launch/patchworkpp.launch.py), message after message.  Prints, per message,
// the sizes and checksums of the three payloads the node would publish (FNV-1a of the cloud, the sum of the points' FNV-1a for the two lists); tests/test_gpu_parity.py::test_ros_node_core
// compares them with what the C-ABI gives for the same sequence.
//
//   g++ -std=c++17 -O2 -I ../ros/include -I ../include -I ../../include ros_core_demo.cpp -L ../lib -lpwpp_hip -o ros_core_demo
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "patchworkpp_ros/segmentation_core.hpp"

static std::vector<float> read_bin(const std::string &path) {  // float32 x, y, z, intensity records
    std::vector<float> v;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return v;
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)bytes / sizeof(float));
    if (std::fread(v.data(), sizeof(float), v.size(), f) != v.size()) v.clear();
    std::fclose(f);
    v.resize(v.size() / 4 * 4);
    return v;
}
static unsigned long long fnv1a(const uint8_t *d, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ d[i]) * 1099511628211ull;
    return h;
}
static unsigned long long fnv1a(const std::vector<uint8_t> &d) { return fnv1a(d.data(), d.size()); }
// the order inside a patch's part of a list is the scatter order (atomics: it differs from run to run): the lists are compared
// as multisets of 16-byte points -- the sum of the points' checksums
static unsigned long long point_sum(const std::vector<uint8_t> &d) {
    unsigned long long s = 0;
    for (size_t i = 0; i + 16 <= d.size(); i += 16) s += fnv1a(d.data() + i, 16);
    return s;
}

int main(int argc, char **argv) {
    // the launch file's parameters, as the node would receive them from `ros2 launch`
    const std::map<std::string, double> dbl = {{"sensor_height", 1.88}, {"th_seeds", 0.3}, {"th_dist", 0.125}, {"th_seeds_v", 0.25},
                                               {"th_dist_v", 0.9}, {"max_range", 80.0}, {"min_range", 1.0}, {"uprightness_thr", 0.101}};
    const std::map<std::string, int> ints = {{"num_iter", 3}, {"num_lpr", 20}, {"num_min_pts", 0}};
    try {
        const patchwork::Params params = patchworkpp_ros::declare_parameters(
            [&](const std::string &n, double d) { return dbl.count(n) ? dbl.at(n) : d; },
            [&](const std::string &n, int d) { return ints.count(n) ? ints.at(n) : d; },
            [&](const std::string &, bool) { return false; });
        patchworkpp_ros::SegmentationCore core(params);
        const std::vector<patchworkpp_ros::Field> fields = {{"intensity", 0, patchworkpp_ros::kFloat32, 1}, {"x", 4, patchworkpp_ros::kFloat32, 1},
                                                            {"ring", 8, 4 /* UINT16 */, 1},              {"y", 12, patchworkpp_ros::kFloat32, 1},
                                                            {"z", 20, patchworkpp_ros::kFloat32, 1},     {"time", 24, 8 /* FLOAT64 */, 1}};
        // --layout=odd: a message whose step and offsets are NOT multiples of four (the core repacks it); --layout=bad: a field that
        // reaches beyond point_step (the core must refuse it before anything reads past a point)
        std::vector<patchworkpp_ros::Field> alt = fields;
        uint32_t step = 32;
        int first = 1;
        if (argc > 1 && std::string(argv[1]) == "--layout=odd") {
            alt = {{"x", 1, patchworkpp_ros::kFloat32, 1}, {"y", 9, patchworkpp_ros::kFloat32, 1}, {"z", 17, patchworkpp_ros::kFloat32, 1}};
            step = 29;
            first = 2;
        } else if (argc > 1 && std::string(argv[1]) == "--layout=bad") {
            alt = {{"x", 4, patchworkpp_ros::kFloat32, 1}, {"y", 12, patchworkpp_ros::kFloat32, 1}, {"z", 30, patchworkpp_ros::kFloat32, 1}};
            first = 2;
        }
        for (int a = first; a < argc; ++a) {
            const std::vector<float> pts = read_bin(argv[a]);
            const size_t n = pts.size() / 4;
            std::vector<uint8_t> blob(n * step + 8, 0xA5);  // (whatever lies between the fields must not matter)
            for (size_t i = 0; i < n; ++i) {
                if (alt.size() == fields.size()) std::memcpy(&blob[i * step + 0], &pts[i * 4 + 3], 4);
                for (const auto &f : alt) {
                    const int c = f.name == "x" ? 0 : (f.name == "y" ? 1 : (f.name == "z" ? 2 : -1));
                    if (c >= 0 && f.offset + 4 <= step) std::memcpy(&blob[i * step + f.offset], &pts[i * 4 + c], 4);
                }
            }
            patchworkpp_ros::CloudView msg;
            msg.height = 1;
            msg.width = (uint32_t)n;
            msg.point_step = step;
            msg.fields = alt.data();
            msg.num_fields = alt.size();
            msg.data = blob.data();
            msg.data_size = n * step;
            const patchworkpp_ros::SegmentationCore::Output out = core.estimate(msg);
            std::printf("{\"file\": \"%s\", \"points\": %zu, \"cloud\": [%u, %u, \"%016llx\"], \"ground\": [%u, %u, \"%016llx\"], "
                        "\"nonground\": [%u, %u, \"%016llx\"], \"time_us\": %.1f}\n",
                        argv[a], n, out.cloud.width, out.cloud.point_step, fnv1a(out.cloud.data), out.ground.width, out.ground.point_step,
                        point_sum(out.ground.data), out.nonground.width, out.nonground.point_step, point_sum(out.nonground.data), out.time_taken_us);
        }
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
