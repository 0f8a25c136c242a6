// TEST INFRASTRUCTURE: C-ABI driver around the reference's dataset-preprocessing grid subsampling (features mean + label
// majority vote), tensorflow/ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106 — the core the CPython
// module `grid_subsampling.compute` wraps (wrapper.cpp:58-286; the wrapper itself does not build against numpy 2).
#include <vector>
#include "/root/reference/tensorflow/ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.h"

extern "C" int ref_grid_subsampling_full(int n, const float* points, int fdim, const float* features, int ldim, const int* classes, float dl,
                                         float* out_points, float* out_features, int* out_classes, int cap)
{
    std::vector<PointXYZ> p((size_t)n), sub;
    for (int i = 0; i < n; i++) p[i] = PointXYZ(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    std::vector<float> of(features, features + (size_t)n * fdim), sf;
    std::vector<int> oc(classes, classes + (size_t)n * ldim), sc;
    grid_subsampling(p, sub, of, sf, oc, sc, dl, 0);
    const int m = (int)sub.size();
    if (m <= cap) {
        for (int i = 0; i < m; i++) { out_points[3 * i] = sub[i].x; out_points[3 * i + 1] = sub[i].y; out_points[3 * i + 2] = sub[i].z; }
        for (size_t i = 0; i < sf.size(); i++) out_features[i] = sf[i];
        for (size_t i = 0; i < sc.size(); i++) out_classes[i] = sc[i];
    }
    return m;
}
