// stand-in for <pcl/point_types.h> — test infrastructure only (oracle/_ref).  The reference's PCL overloads of
// VoxelCalculator are dead code; they only have to compile.
#pragma once
#include <Eigen/Core>
namespace pcl {
struct PointXYZI {
    float x = 0, y = 0, z = 0, intensity = 0;
    Eigen::Vector3f getVector3fMap() const { return Eigen::Vector3f(x, y, z); }
};
}  // namespace pcl
