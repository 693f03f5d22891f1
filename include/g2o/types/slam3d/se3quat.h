// forwards to the MI355X-native facade: see include/plba_g2o/types_slam3d.h (replaces g2o/types/slam3d/se3quat.h of the third-party g2o)
#pragma once
#include "plba_g2o/types_slam3d.h"
