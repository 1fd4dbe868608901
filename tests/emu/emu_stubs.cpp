// CPU test build only: the MFMA ResNet evaluator does not exist without a GPU.
#define GAZ_HOST_EMU 1
#include "../../grok_alpha_zero_amd/csrc/evaluator.hpp"
namespace gaz {
Evaluator* make_resnet_evaluator(const gaz_engine_config&, int, int, int, int, std::string* err) {
    *err = "the ResNet evaluator needs the HIP build";
    return nullptr;
}
bool launch_wave_trunk_c4(hipStream_t, const void*, int, int, const void*) { return false; }
}
