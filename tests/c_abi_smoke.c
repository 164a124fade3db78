/* Links against libmpe_hip.so through include/mpe.h as a plain C program (no compute calls:
 * runs on machines without a GPU). */
#include <stdio.h>
#include <string.h>

#include "mpe.h"

int main(void) {
  mpe_params p;
  mpe_default_params(&p);
  if (p.threshold_value != 140 || p.gaussian_sigma != 0.6 || p.back_projection_pixel_tolerance != 5.0 ||
      p.nearest_neighbour_pixel_tolerance != 7.0 || p.roi_border_thickness != 20)
    return 2;
  printf("%s devices=%d sizeof(result)=%zu sizeof(detections)=%zu\n", mpe_version(), mpe_device_count(),
         sizeof(mpe_result), sizeof(mpe_detections));
  mpe_handle* h = 0;
  int rc = mpe_create(&h, -1);
  if (mpe_device_count() == 0) {
    if (rc != MPE_ERR_NO_DEVICE || h != 0) return 3; /* no CPU fallback */
  } else {
    if (rc != MPE_OK) return 4;
    mpe_destroy(h);
  }
  return 0;
}
