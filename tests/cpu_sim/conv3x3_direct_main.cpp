// host run of conv3x3_direct_kernel<VEC> (musev_amd/csrc/elementwise.hip) for tests/test_kernel_cpu_sim.py
//   argv: in.bin out.bin cin cout n_img h w stride act      in.bin = x | w (packed [cout][9*cin]) | bias, all fp16
#include <stdio.h>
#include "hip_cpu_sim.h"
#define dsw_DECL half_t* dsw = reinterpret_cast<half_t*>(sim_smem)
#include "conv3x3_direct_extract.inc"

int main(int argc, char** argv) {
    if (argc != 10) return 2;
    const int cin = atoi(argv[3]), cout = atoi(argv[4]), h = atoi(argv[6]), w = atoi(argv[7]), stride = atoi(argv[8]), act = atoi(argv[9]);
    const long n_img = atol(argv[5]);
    const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
    const size_t nx = (size_t)n_img * h * w * cin, nw = (size_t)cout * 9 * cin, ny = (size_t)n_img * ho * wo * cout;
    std::vector<half_t> x(nx), wt(nw), bias(cout), y(ny);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(x.data(), 2, nx, f) != nx || fread(wt.data(), 2, nw, f) != nw || fread(bias.data(), 2, cout, f) != (size_t)cout) return 3;
    fclose(f);
    const unsigned blocks = (unsigned)((n_img * ho * wo + 255) / 256);
    const size_t smem = (size_t)8 * 9 * cin * sizeof(half_t);
    if (cin % 8 == 0)
        sim_launch(blocks, cout / 8, 256, smem, [&]() { conv3x3_direct_kernel<true>(x.data(), cin, wt.data(), bias.data(), y.data(), cout, n_img, h, w, ho, wo, stride, act); });
    else
        sim_launch(blocks, cout / 8, 256, smem, [&]() { conv3x3_direct_kernel<false>(x.data(), cin, wt.data(), bias.data(), y.data(), cout, n_img, h, w, ho, wo, stride, act); });
    f = fopen(argv[2], "wb");
    fwrite(y.data(), 2, ny, f);
    fclose(f);
    return 0;
}
