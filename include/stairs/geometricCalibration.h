/* geometricCalibration.h — same name as the reference's header, so that its sources include this build unchanged:
 * stairs::GeometricCalibration::load (reference geometricCalibration.h:32-37).  Forwards to stairs_api.h. */
#ifndef SSD_COMPAT_GEOMETRICCALIBRATION_H_
#define SSD_COMPAT_GEOMETRICCALIBRATION_H_
#include "stairs_api.h"
#endif
