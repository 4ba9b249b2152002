/* window.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::Window (reference window.h:41-53).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_WINDOW_H_
#define SSD_COMPAT_WINDOW_H_
#include "stairs_api.h"
#endif
