/* camera.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::Camera, Camera::DepthFrame, Camera::Frameset (reference camera.h:31-78).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_CAMERA_H_
#define SSD_COMPAT_CAMERA_H_
#include "stairs_api.h"
#endif
