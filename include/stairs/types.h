/* types.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::Point2, Point3, Point3f, Quadrilateral_t (reference types.h:30-115).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_TYPES_H_
#define SSD_COMPAT_TYPES_H_
#include "stairs_api.h"
#endif
