/* transformation.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::GeometricTransformation, CameraToWorld, WorldToCamera, ToExternalWorld (reference transformation.h:79-126).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_TRANSFORMATION_H_
#define SSD_COMPAT_TRANSFORMATION_H_
#include "stairs_api.h"
#endif
