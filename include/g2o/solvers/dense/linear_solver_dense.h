// forwards to the MI355X-native facade: see include/plba_g2o/g2o_compat.h (replaces g2o/solvers/dense/linear_solver_dense.h of the third-party g2o)
#pragma once
#include "plba_g2o/g2o_compat.h"
