// MINIMAL STAND-IN for <open3d/Open3D.h> — only the members of open3d::geometry::PointCloud that the reference-side binding of
// INTEGRATION.md section B touches (points_, normals_: contiguous std::vector<Eigen::Vector3d>, as upstream).  Test infrastructure.
#pragma once
#include <Eigen/Dense>
#include <memory>
#include <vector>

namespace open3d {
namespace geometry {
class PointCloud {
public:
    std::vector<Eigen::Vector3d> points_;
    std::vector<Eigen::Vector3d> normals_;
};
}  // namespace geometry
}  // namespace open3d
