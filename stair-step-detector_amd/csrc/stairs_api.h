/*
 * stairs_api.h — the reference's C++ class surface for the per-frame path, kept so that a
 * detect-stairs.cpp-shaped main compiles unchanged against libssd_hip.so:
 *   stairs::Pointcloud              pointcloud.h:32-42      (process() prints Stairs::serialize(), pointcloud.cpp:625)
 *   stairs::Stairs                  stairs.h:30-39
 *   stairs::GeometricTransformation transformation.h:102-126
 *   stairs::Camera::DepthFrame      camera.h:44-63  — here a view of W*H float xyz vertices (the output of
 *                                   rs2::pointcloud::calculate, pointcloud.cpp:138) instead of an rs2::depth_frame
 *   stairs::Window                  window.h — the GL sink; a no-op here (it never affects results)
 * Everything forwards to the C ABI in include/ssd_hip.h; there is no CPU implementation behind it.
 */
#ifndef STAIRS_API_H_
#define STAIRS_API_H_

#include "../../include/ssd_hip.h"
#include <array>
#include <string>
#include <vector>

namespace stairs
{

using Coordinate_t = double;
struct Point2 { Coordinate_t x = 0, y = 0; };
struct Point3 { Coordinate_t x = 0, y = 0, z = 0; };
using Quadrilateral_t = std::array<Point2, 4>;

struct Stairs
{
  struct StairStep
  {
    Coordinate_t height;
    Quadrilateral_t quadrilateral;
  };
  std::vector<StairStep> stairSteps;
  std::string serialize() const;
};

class Window
{
public:
  explicit Window(const char *) {}
  operator bool() const { return true; }
};

class Camera
{
public:
  struct DepthFrame
  {
    const float *vertices;   /* width*height x (x,y,z), row-major, invalid = (0,0,0) */
    int width, height;
  };
};

class GeometricTransformation
{
public:
  using RefPoints = std::array<Point3, 3>;
  GeometricTransformation();                                                        /* identity, transformation.h:51-55 */
  GeometricTransformation(const RefPoints &worldPoints, const RefPoints &cameraPoints);
  explicit GeometricTransformation(const ssd_calibration &constants) : _cal(constants) {}
  const ssd_calibration &constants() const { return _cal; }

private:
  GeometricTransformation(const GeometricTransformation &) = delete;
  ssd_calibration _cal;
};

/* geometricCalibration.h:32-37: only the offline half (load) is on the path's boundary */
class GeometricCalibration
{
public:
  static GeometricTransformation load();      /* reads "calibration-triangle" and "calibration-points" from the working directory */
};

class Pointcloud
{
public:
  Pointcloud(const Window &window, const GeometricTransformation &trans);
  ~Pointcloud();
  void process(const Camera::DepthFrame &frame) const;     /* prints one line to std::cout */
  Stairs detect(const Camera::DepthFrame &frame) const;    /* the Stairs value process() serialises */

private:
  Pointcloud(const Pointcloud &) = delete;
  const Window &_window;
  const GeometricTransformation &_transformation;
  mutable ssd_handle *_handle = nullptr;
  mutable int _width = 0, _height = 0;
};

} // namespace stairs

#endif /* STAIRS_API_H_ */
