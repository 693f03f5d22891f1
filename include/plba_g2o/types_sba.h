// types_sba.h — drop-in for the reference header IMU/types_sba.h (VertexSBAPointXYZ, :40-58).  API surface only.
#pragma once
#include "plba_g2o/g2o_compat.h"

namespace g2o {

class VertexSBAPointXYZ : public BaseVertex<3, Vector3d> {
public:
    VertexSBAPointXYZ() { setToOriginImpl(); }
    void setToOriginImpl() override { for (int i = 0; i < 3; ++i) _estimate[i] = 0.0; }
    void oplusImpl(const double* u) override { for (int i = 0; i < 3; ++i) _estimate[i] += u[i]; }
    int estimateDimension() const override { return 3; }
    bool read(std::istream& is) override { for (int i = 0; i < 3; ++i) is >> _estimate[i]; return true; }          // IMU/types_sba.cpp:36-42
    bool write(std::ostream& os) const override { for (int i = 0; i < 3; ++i) os << _estimate[i] << " "; return os.good(); }   // :44-51
};

}  // namespace g2o
