/* stairs_api.cpp — see stairs_api.h */
#include "../../include/stairs/stairs_api.h"
#include <iostream>
#include <stdexcept>

namespace stairs
{

std::string Stairs::serialize() const
{
  ssd_frame_result r{};
  r.n_steps = static_cast<int>(stairSteps.size() < SSD_MAX_STEPS ? stairSteps.size() : SSD_MAX_STEPS);
  for(int i = 0; i < r.n_steps; i++)
  {
    r.steps[i].height = stairSteps[i].height;
    for(int k = 0; k < 4; k++)
    {
      r.steps[i].quad[2 * k] = stairSteps[i].quadrilateral[k].x;
      r.steps[i].quad[2 * k + 1] = stairSteps[i].quadrilateral[k].y;
    }
  }
  std::string buf(SSD_LINE_CAP, '\0');
  const int n = ssd_serialize(&r, buf.data(), buf.size());
  buf.resize(n > 0 ? n : 0);
  return buf;
}

GeometricTransformation::GeometricTransformation()
{
  ssd_calibration_identity(&_cal);
}

GeometricTransformation::GeometricTransformation(const RefPoints &w, const RefPoints &c)
{
  const double wp[9] = { w[0].x, w[0].y, w[0].z, w[1].x, w[1].y, w[1].z, w[2].x, w[2].y, w[2].z };
  const double cp[9] = { c[0].x, c[0].y, c[0].z, c[1].x, c[1].y, c[1].z, c[2].x, c[2].y, c[2].z };
  if(ssd_calibration_from_points(wp, cp, &_cal) != SSD_OK)
    throw std::invalid_argument(ssd_last_error());
}

/* Transformation_<3>::transform (transformation.h:59-64): a * x + b, row sums left to right, then the translation */
Point3 CameraToWorld::apply(double x, double y, double z) const
{
  const double *a = _camera.a, *b = _camera.b;
  Point3 w;
  w.x = (a[0] * x + a[1] * y) + a[2] * z;
  w.y = (a[3] * x + a[4] * y) + a[5] * z;
  w.z = (a[6] * x + a[7] * y) + a[8] * z;
  w.x = w.x + b[0];
  w.y = w.y + b[1];
  w.z = w.z + b[2];
  return w;
}

/* Transformation_<3>::transformInv (transformation.h:66-69): aInv * (x - b); for the camera transformation aInv is
 * exactly the transpose of a (transformation.cpp:139-142) */
Point3 WorldToCamera::operator()(const Point3 &p) const
{
  const double *a = _camera.a, *b = _camera.b;
  const double dx = p.x - b[0], dy = p.y - b[1], dz = p.z - b[2];
  Point3 c;
  c.x = (a[0] * dx + a[3] * dy) + a[6] * dz;
  c.y = (a[1] * dx + a[4] * dy) + a[7] * dz;
  c.z = (a[2] * dx + a[5] * dy) + a[8] * dz;
  return c;
}

/* ToExternalWorld::operator() (transformation.cpp:190-194): 2-D rotation + translation of (x, y); z = worldZ + p.z */
Point3 ToExternalWorld::operator()(const Point3 &p) const
{
  const double *r = _world.r2, *t = _world.t2;
  double ex = r[0] * p.x + r[1] * p.y;
  double ey = r[2] * p.x + r[3] * p.y;
  ex = ex + t[0];
  ey = ey + t[1];
  return Point3{ ex, ey, _world.world_z + p.z };
}

GeometricTransformation GeometricCalibration::load()
{
  ssd_calibration cal;
  int loaded = 0;
  ssd_calibration_load("calibration-triangle", "calibration-points", &cal, &loaded, nullptr, nullptr);
  (void)loaded;      /* the reference logs "calibration points could not be loaded" and carries on with identity (:199-202) */
  return GeometricTransformation(cal);
}

Pointcloud::Pointcloud(const Window &window, const GeometricTransformation &trans)
: _window(window),
  _transformation(trans)
{
}

Pointcloud::~Pointcloud()
{
  ssd_destroy(_handle);
}

Stairs Pointcloud::detect(const Camera::DepthFrame &frame) const
{
  if(!_handle || frame.width != _width || frame.height != _height)
  {
    ssd_destroy(_handle);
    _handle = nullptr;
    ssd_config cfg;
    if(ssd_default_config(&cfg, frame.width, frame.height) != SSD_OK)
      throw std::runtime_error(ssd_last_error());
    cfg.max_frames_per_batch = 1;
    if(ssd_create(&cfg, &_transformation.constants(), 0, &_handle) != SSD_OK)
      throw std::runtime_error(ssd_last_error());
    _width = frame.width;
    _height = frame.height;
  }
  ssd_frame_result r;
  if(ssd_process_host(_handle, frame.vertices, 1, &r) != SSD_OK)
    throw std::runtime_error(ssd_last_error());
  if(r.status & SSD_ST_THROW)
    throw std::invalid_argument("Quadrilateral is not usable (quadrilateralTest.cpp:283-372)");   /* as the reference does */
  Stairs s;
  s.stairSteps.resize(r.n_steps);
  for(int i = 0; i < r.n_steps; i++)
  {
    s.stairSteps[i].height = r.steps[i].height;
    for(int k = 0; k < 4; k++)
      s.stairSteps[i].quadrilateral[k] = Point2{ r.steps[i].quad[2 * k], r.steps[i].quad[2 * k + 1] };
  }
  return s;
}

void Pointcloud::process(const Camera::DepthFrame &frame) const
{
  std::cout << detect(frame).serialize() << std::endl;     /* pointcloud.cpp:625 */
}

} // namespace stairs
