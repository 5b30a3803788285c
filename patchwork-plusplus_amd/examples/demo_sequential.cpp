// demo_sequential.cpp -- console twin of the reference's cpp/patchworkpp/examples/demo_sequential.cpp
// (one long-lived PatchWorkpp object over a directory of KITTI .bin frames, in sorted order),
// without the Open3D window: prints what the reference prints plus a checksum of the index sets.
// Built against patchwork-plusplus_amd/include/patchwork/patchworkpp.h (no Eigen needed).
//
//   g++ -std=c++17 -O2 -I ../include -I ../../include demo_sequential.cpp -L ../lib -lpwpp_hip \
//       -Wl,-rpath,'$ORIGIN/../lib' -o demo_sequential
//   ./demo_sequential frame0.bin frame1.bin ...
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#include "patchwork/patchworkpp.h"

// reference cpp/patchworkpp/examples/demo_visualize.cpp:18-34 (float32 x,y,z,intensity records)
static std::vector<float> read_bin(const std::string &path) {
    std::vector<float> v;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        std::printf("Could not open %s\n", path.c_str());
        return v;
    }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)bytes / sizeof(float));
    if (std::fread(v.data(), sizeof(float), v.size(), f) != v.size()) v.clear();
    std::fclose(f);
    v.resize(v.size() / 4 * 4);
    return v;
}

int main(int argc, char **argv) {
    std::vector<std::string> files(argv + 1, argv + argc);
    std::sort(files.begin(), files.end());
    patchwork::Params patchwork_parameters;
    patchwork_parameters.verbose = false;
    try {
        patchwork::PatchWorkpp Patchworkpp(patchwork_parameters);
        for (const std::string &path : files) {
            const std::vector<float> cloud = read_bin(path);
            if (cloud.empty()) continue;
            Patchworkpp.estimateGround(cloud.data(), (int)(cloud.size() / 4), 4);
            const patchwork::Indices g = Patchworkpp.getGroundIndices();
            const patchwork::Indices ng = Patchworkpp.getNongroundIndices();
            const patchwork::Cloud ground = Patchworkpp.getGround();
            const patchwork::Cloud normals = Patchworkpp.getNormals();
            long long sum = 0;
            for (int i = 0; i < g.rows(); ++i) sum += g(i);
            std::printf("%s: Origianl Points #: %zu  Ground Points #: %d  Nonground Points #: %d  patches: %d  "
                        "idxsum: %lld  height: %.6f  time: %.1f us\n",
                        path.c_str(), cloud.size() / 4, ground.rows(), ng.rows(), normals.rows(), sum,
                        Patchworkpp.getHeight(), Patchworkpp.getTimeTaken());
        }
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
