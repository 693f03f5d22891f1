// forwards to the MI355X-native facade (replaces g2o/solvers/eigen/linear_solver_eigen.h)
#pragma once
#include "plba_g2o/g2o_compat.h"
