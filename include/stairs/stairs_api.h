/*
 * stairs_api.h — the reference's C++ class surface for the per-frame path, kept so that the reference's own
 * detect-stairs.cpp compiles UNCHANGED against this build (tests/test_cxx_surface.py compiles it in place):
 *
 *   stairs::Pointcloud              pointcloud.h:32-42      (process() prints Stairs::serialize(), pointcloud.cpp:625)
 *   stairs::Stairs                  stairs.h:30-39
 *   stairs::GeometricTransformation transformation.h:102-126, with cameraToWorld() / worldToCamera() /
 *                                   toExternalWorld() (transformation.h:79-100, 117-119)
 *   stairs::GeometricCalibration    geometricCalibration.h:32-37 — load() only (geometricCalibration.cpp:185-203)
 *   stairs::Camera                  camera.h:31-78 — start() / waitForFrames() / Frameset::depthFrame() over this
 *                                   build's frame source instead of a RealSense pipeline; DepthFrame is a view of
 *                                   W*H float xyz vertices (the output of rs2::pointcloud::calculate, pointcloud.cpp:138)
 *   stairs::Window                  window.h:41-53 — the GL sink; draws nothing (it never affects results); converts
 *                                   to false when the frame source is exhausted, as the reference's does when closed
 *
 * The headers next to this one carry the reference's file names (window.h, geometricCalibration.h, pointcloud.h, ...)
 * and forward here: compile with -I include/stairs.  Pointcloud / Stairs / GeometricTransformation /
 * GeometricCalibration live in libssd_hip.so and forward to its C ABI (include/ssd_hip.h) — there is no CPU
 * implementation behind them; Camera and Window::operator bool live in libssd_source.so (include/ssd_source.h).
 */
#ifndef STAIRS_API_H_
#define STAIRS_API_H_

#include "../ssd_hip.h"
#include <array>
#include <memory>
#include <string>
#include <vector>

namespace stairs
{

/* types.h:30-115 */
using Coordinate_t = double;
struct Point3f { float x, y, z; };
struct Point3;
struct Point2
{
  Coordinate_t x = 0, y = 0;
  Point2() = default;
  Point2(Coordinate_t x_, Coordinate_t y_) : x(x_), y(y_) {}
  Point2(const Point3 &p);
};
struct Point3
{
  Coordinate_t x = 0, y = 0, z = 0;
  Point3() = default;
  Point3(Coordinate_t x_, Coordinate_t y_, Coordinate_t z_) : x(x_), y(y_), z(z_) {}
  Point3(const Point2 &p, Coordinate_t z_) : x(p.x), y(p.y), z(z_) {}
};
inline Point2::Point2(const Point3 &p) : x(p.x), y(p.y) {}
using Quadrilateral_t = std::array<Point2, 4>;

/* stairs.h:30-39 */
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

/* camera.h:31-78 */
class Camera
{
public:
  struct DepthFrame
  {
    const float *vertices;   /* width*height x (x,y,z), row-major, invalid = (0,0,0) */
    int width, height;
    std::shared_ptr<const void> keep = nullptr;     /* whoever owns the vertices (a Frameset's frames own theirs) */
  };
  struct Frameset
  {
    DepthFrame depthFrame() const { return depth; }
    DepthFrame depth;
  };

  /* camera.cpp:27-44: opens the frame source.  Configured by the environment, because the reference's main takes no
   * arguments: SSD_SOURCE_FILE (raw float32 xyz frames; else synthetic staircases), SSD_SOURCE_WIDTH / _HEIGHT (640 x 480,
   * the reference's resolution, configuration.h:36), SSD_SOURCE_FRAMES (1), SSD_SOURCE_STEPS (3), SSD_SOURCE_SEED (12345) */
  void start();
  /* camera.cpp:46-49: the next frame; throws std::runtime_error when the source is exhausted (rs2 throws on a timeout) */
  Frameset waitForFrames();

private:
  struct Source;
  std::shared_ptr<Source> _source;
};

/* window.h:41-53 */
class Window
{
public:
  explicit Window(const char *) {}
  /* example.hpp's window::operator bool: false once the window was closed — here: once every frame of the source
   * a Camera opened has been handed out (true while no source was opened) */
  operator bool() const;
  void show(const Camera::Frameset &) {}
  void setViewport(int) const {}
  void waitClose() {}
};

/* transformation.h:79-100: functors over the constants of a GeometricTransformation; same operation order as the
 * reference (and as the kernels): row sums left to right, then the translation; evaluated inside the library, which
 * is compiled without FMA contraction */
struct CameraToWorld
{
  template<typename SrcPointType>
  Point3 operator()(const SrcPointType &p) const { return apply(static_cast<double>(p.x), static_cast<double>(p.y), static_cast<double>(p.z)); }
  Point3 apply(double x, double y, double z) const;
  const ssd_calibration &_camera;
};
struct WorldToCamera
{
  Point3 operator()(const Point3 &p) const;        /* transformInv: transposed(a) * (p - b), transformation.h:66-69, .cpp:139-142 */
  const ssd_calibration &_camera;
};
struct ToExternalWorld
{
  Point3 operator()(const Point3 &p) const;        /* transformation.cpp:190-194 */
  const ssd_calibration &_world;
};

class GeometricTransformation
{
public:
  using RefPoints = std::array<Point3, 3>;
  GeometricTransformation();                                                        /* identity, transformation.h:51-55 */
  GeometricTransformation(const RefPoints &worldPoints, const RefPoints &cameraPoints);
  explicit GeometricTransformation(const ssd_calibration &constants) : _cal(constants) {}
  const CameraToWorld &cameraToWorld() const { return _cameraToWorld; }
  const WorldToCamera &worldToCamera() const { return _worldToCamera; }
  const ToExternalWorld &toExternalWorld() const { return _toExternalWorld; }
  const ssd_calibration &constants() const { return _cal; }        /* what ssd_create takes */

private:
  GeometricTransformation(const GeometricTransformation &) = delete;
  ssd_calibration _cal;
  const CameraToWorld _cameraToWorld{ _cal };
  const WorldToCamera _worldToCamera{ _cal };
  const ToExternalWorld _toExternalWorld{ _cal };
};

/* geometricCalibration.h:32-37: only the offline half (load) is on the path's boundary */
class GeometricCalibration
{
public:
  static GeometricTransformation load();      /* reads "calibration-triangle" and "calibration-points" from the working directory */
};

/* pointcloud.h:32-42 */
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
