// resnet.hip — placeholder until the MFMA evaluator lands (next commit).
#include "evaluator.hpp"
namespace gaz {
Evaluator* make_resnet_evaluator(const gaz_engine_config&, int, int, int, int, std::string* err) {
    *err = "ResNet evaluator not built yet";
    return nullptr;
}
}
