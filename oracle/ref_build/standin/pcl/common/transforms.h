// stand-in for <pcl/common/transforms.h> — included by the reference's voxel_calculator.hpp, never used.
#pragma once
