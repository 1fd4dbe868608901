// CPU test build only: the MFMA ResNet evaluator does not exist without a GPU.
#define GAZ_HOST_EMU 1
#include "../../grok_alpha_zero_amd/csrc/evaluator.hpp"
namespace gaz {
Evaluator* make_resnet_evaluator(const gaz_engine_config&, int, int, int, int, std::string* err) {
    *err = "the ResNet evaluator needs the HIP build";
    return nullptr;
}
bool launch_wave_trunk_c4(hipStream_t, const void*, int, int, const void*) { return false; }
bool launch_wave_trunk_c4_gumbel(hipStream_t, const void*, int, int, const void*) { return false; }
bool launch_wave_trunk_gmk(hipStream_t, const void*, int, int, const void*) { return false; }
}

// test hook (tests/test_tile_perm.py): the host-side tile permutation of the trunk kernel's edge tiles
#include "../../grok_alpha_zero_amd/csrc/tile_perm.hpp"
extern "C" int gaz_test_tile_perm(int H, int W, int boards, int rows, int wave_rows, int per_wave_row, uint8_t* perm_out, unsigned* sitout_out,
                                  int* boff_out, int* clashes_out) {
    const gaz::TileLayout L = gaz::tile_layout(H, W, boards, rows, wave_rows, per_wave_row);
    if (L.perm.empty()) return 0;
    for (size_t i = 0; i < L.perm.size(); ++i) perm_out[i] = L.perm[i];
    for (int t = 0; t < rows / 16; ++t) sitout_out[t] = gaz::tile_sitout(L.perm, L.boff, H, W, t);
    for (size_t b = 0; b < L.boff.size() && boff_out; ++b) boff_out[b] = L.boff[b];
    if (clashes_out) *clashes_out = L.clashes;
    return (int)L.perm.size();
}
