// drop-in for the reference header IMU/IMUPreintegrator.h: same class names and methods, HIP-backed (see g2o_compat.h)
#pragma once
#include "plba_g2o/g2o_compat.h"
