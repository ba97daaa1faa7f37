// stand-in for <yaml-cpp/yaml.h> — included by the reference's map_eval.h; map_eval.cpp and voxel_calculator.cpp use nothing
// of it (only map_eval_main.cpp, which oracle/_ref does not compile, does).  Test infrastructure only.
#pragma once
