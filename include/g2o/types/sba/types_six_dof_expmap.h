// forwards to the MI355X-native facade: see include/plba_g2o/types_six_dof_expmap.h (replaces g2o/types/sba/types_six_dof_expmap.h of the third-party g2o)
#pragma once
#include "plba_g2o/types_six_dof_expmap.h"
