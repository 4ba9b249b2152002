/* stairs.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::Stairs (reference stairs.h:30-39).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_STAIRS_H_
#define SSD_COMPAT_STAIRS_H_
#include "stairs_api.h"
#endif
