// demo_eigen.cpp -- the calls of the reference's cpp/patchworkpp/examples/demo_visualize.cpp:70-93 with
// the reference's own types: Eigen::MatrixXf in, Eigen::MatrixX3f / Eigen::VectorXi out (no Open3D).
// Needs Eigen on the include path (the class mirror enables these overloads with __has_include):
//   g++ -std=c++17 -O2 -I <eigen> -I ../include -I ../../include demo_eigen.cpp -L ../lib -lpwpp_hip \
//       -Wl,-rpath,'$ORIGIN/../lib' -o demo_eigen && ./demo_eigen 000000.bin
#include <Eigen/Dense>
#include <cstdio>
#include <vector>
#include "patchwork/patchworkpp.h"
int main(int argc, char **argv) {
    patchwork::Params patchwork_parameters;
    patchwork_parameters.verbose = false;
    try {
        patchwork::PatchWorkpp Patchworkpp(patchwork_parameters);
        std::vector<float> raw;
        FILE *f = argc > 1 ? std::fopen(argv[1], "rb") : nullptr;
        if (!f) { std::printf("no input\n"); return 2; }
        float rec[4];
        while (std::fread(rec, sizeof(float), 4, f) == 4) raw.insert(raw.end(), rec, rec + 4);
        std::fclose(f);
        const int n = (int)(raw.size() / 4);
        Eigen::MatrixXf cloud(n, 4);
        for (int i = 0; i < n; ++i) for (int c = 0; c < 4; ++c) cloud(i, c) = raw[(size_t)i * 4 + c];
        Patchworkpp.estimateGround(cloud);
        Eigen::MatrixX3f ground = Patchworkpp.getGround();
        Eigen::MatrixX3f nonground = Patchworkpp.getNonground();
        Eigen::VectorXi ground_idx = Patchworkpp.getGroundIndices();
        Eigen::VectorXi nonground_idx = Patchworkpp.getNongroundIndices();
        Eigen::MatrixX3f centers = Patchworkpp.getCenters();
        Eigen::MatrixX3f normals = Patchworkpp.getNormals();
        long long s = 0; for (int i = 0; i < ground_idx.rows(); ++i) s += ground_idx(i);
        bool aligned = true;
        for (int i = 0; i < ground.rows(); ++i) aligned = aligned && ground(i, 2) == cloud(ground_idx(i), 2);
        std::printf("Origianl Points  #: %d\nGround Points    #: %d\nNonground Points #: %d\npatches: %d idxsum: %lld aligned: %d\n",
                    (int)cloud.rows(), (int)ground.rows(), (int)nonground.rows(), (int)centers.rows(), s, aligned ? 1 : 0);
        // expressions that only compile when the getters return real Eigen objects (proxies did not)
        auto nrm = Patchworkpp.getNormals();
        const float nz0 = nrm.col(2)(0);
        const float x0 = Patchworkpp.getGround().row(0)(0);
        const auto gt = Patchworkpp.getGround().transpose();
        std::printf("eigen expressions: nz0 %s normals(0,2), x0 %s ground(0,0), transpose %dx%d\n", nz0 == normals(0, 2) ? "==" : "!=",
                    x0 == ground(0, 0) ? "==" : "!=", (int)gt.rows(), (int)gt.cols());
        patchwork_parameters.verbose = true;  // the reference's "Time taken : ..." line (patchworkpp.cpp:323-333)
        patchwork::PatchWorkpp verbose_one(patchwork_parameters);
        verbose_one.estimateGround(cloud);
        (void)nonground_idx;
    } catch (const std::exception &e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
