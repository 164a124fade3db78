// marker_yaml.h — reader for the reference's marker file format (private parameter `marker_positions`,
// monocular_pose_estimator/src/monocular_pose_estimator.cpp:63-83; file layout of
// monocular_pose_estimator/marker_positions/demo_marker_positions.yaml):
//   marker_positions:
//     - x: 0.0714197
//       y: 0.0800214
//       z: 0.0622611
#ifndef MPE_COMPAT_MARKER_YAML_H_
#define MPE_COMPAT_MARKER_YAML_H_

#include "monocular_pose_estimator_lib/facade_namespace.h"
#include <cstdlib>
#include <fstream>
#include <string>

#include "monocular_pose_estimator_lib/datatypes.h"

MPE_FACADE_BEGIN

inline bool read_markers(const char* path, List4DPoints& out) {
  std::ifstream in(path);
  if (!in) return false;
  std::string line;
  double cur[3] = {0, 0, 0};
  int have = 0;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    for (int k = 0; k < 3; ++k) {
      const char key[3] = {"xyz"[k], ':', 0};
      const size_t p = line.find(key);
      if (p == std::string::npos) continue;
      cur[k] = std::atof(line.c_str() + p + 2);
      have |= 1 << k;
    }
    if (have == 7) {
      Vector4d v;
      v(0) = cur[0];
      v(1) = cur[1];
      v(2) = cur[2];
      v(3) = 1.0;
      out.push_back(v);
      have = 0;
    }
  }
  return !out.empty();
}


MPE_FACADE_END  // namespace monocular_pose_estimator
#endif
