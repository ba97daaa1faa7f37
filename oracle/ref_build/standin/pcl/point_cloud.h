// stand-in for <pcl/point_cloud.h> — test infrastructure only (oracle/_ref).
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT>
class PointCloud {
  public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    std::vector<PointT> points;
};
}  // namespace pcl
