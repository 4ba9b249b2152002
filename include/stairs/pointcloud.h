/* pointcloud.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::Pointcloud (reference pointcloud.h:32-42).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_POINTCLOUD_H_
#define SSD_COMPAT_POINTCLOUD_H_
#include "stairs_api.h"
#endif
