// Prints the matrices of the PRODUCT's host math (renderer_amd/csrc/rnd_math.hpp) for the given camera,
// as hex floats, so the test can compare them bit for bit with the oracle's (orc_get_matrices).
#include <cstdio>
#include <cstdlib>

#include "../renderer_amd/csrc/rnd_math.hpp"

int main(int argc, char** argv) {
    if (argc != 21) return 2;
    float t[16];
    for (int i = 0; i < 16; ++i) t[i] = strtof(argv[1 + i], nullptr);
    float vfov = strtof(argv[17], nullptr), aspect = strtof(argv[18], nullptr), zn = strtof(argv[19], nullptr),
          zf = strtof(argv[20], nullptr);
    rnd::Mat4 cam = rnd::Mat4::from_cols_array(t);
    rnd::Mat4 proj = rnd::perspective_lh(vfov, aspect, zn, zf);
    rnd::Mat4 view = rnd::affine_inverse(cam);
    rnd::Mat4 pv = proj * view;
    rnd::Mat4 inv = rnd::inverse(pv);
    rnd::Mat4 look = rnd::look_to_lh({t[12], t[13], t[14]}, {t[8], t[9], t[10]}, {0, 1, 0});
    rnd::Mat4 ortho = rnd::orthographic_lh(-vfov, aspect, -zn, zf, zn, zf);
    const rnd::Mat4* ms[6] = {&proj, &view, &pv, &inv, &look, &ortho};
    for (auto m : ms) {
        float a[16];
        m->to_cols_array(a);
        for (int i = 0; i < 16; ++i) printf("%a ", a[i]);
        printf("\n");
    }
    // shadow cascades of a directional light pointing along -y + 0.5 z of the camera transform
    rnd::Mat4 cascades[4];
    rnd::shadow_cascade_projviews(inv, {-t[4] + 0.5f * t[8], -t[5] + 0.5f * t[9], -t[6] + 0.5f * t[10]}, cascades);
    for (const auto& m : cascades) {
        float a[16];
        m.to_cols_array(a);
        for (int i = 0; i < 16; ++i) printf("%a ", a[i]);
        printf("\n");
    }
    printf("%a %a %a\n", rnd::round_to_f16(0.02f), rnd::round_to_f16(0.04f), rnd::round_to_f16(0.08f));
    return 0;
}
