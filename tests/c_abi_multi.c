/* mpe_estimate_batch_multi from a plain C host (the stated caller of the library is a C++ host process): the
 * batch sharded over n_dev handles — on n_dev different GPUs when the box has them, else several handles on
 * GPU 0 — must give byte-identical records to one mpe_estimate_batch call.
 *   usage: c_abi_multi frames.raw n_frames rows cols n_dev markers.txt(n x 3 doubles as text)        */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpe.h"

int main(int argc, char** argv) {
  if (argc != 7) return 2;
  const int n = atoi(argv[2]), rows = atoi(argv[3]), cols = atoi(argv[4]), n_dev = atoi(argv[5]);
  const size_t fb = (size_t)rows * cols;
  uint8_t* frames = (uint8_t*)malloc(fb * n);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(frames, 1, fb * n, f) != fb * n) return 3;
  fclose(f);
  double markers[3 * MPE_MAX_MARKERS];
  int nm = 0;
  f = fopen(argv[6], "r");
  if (!f) return 3;
  while (nm < MPE_MAX_MARKERS && fscanf(f, "%lf %lf %lf", &markers[3 * nm], &markers[3 * nm + 1], &markers[3 * nm + 2]) == 3) ++nm;
  fclose(f);
  const double K[9] = {615.652408400557, 0, 362.655454167686, 0, 616.760184718123, 256.67210750994, 0, 0, 1};
  const double D[5] = {-0.358561237166698, 0.149312912580924, 0.000484551782515636, -0.000200189442379448, 0};
  mpe_params p;
  mpe_default_params(&p);
  const int n_gpu = mpe_device_count();
  if (n_gpu < 1) return 4;
  mpe_handle* hs[16];
  for (int d = 0; d < n_dev; ++d)
    if (mpe_create(&hs[d], d % n_gpu) != MPE_OK) return 5;
  mpe_result* one = (mpe_result*)calloc(n, sizeof(mpe_result));
  mpe_result* many = (mpe_result*)calloc(n, sizeof(mpe_result));
  if (mpe_estimate_batch(hs[0], frames, n, rows, cols, cols, fb, 0, markers, nm, K, D, 5, &p, one) != MPE_OK) {
    fprintf(stderr, "single: %s\n", mpe_last_error(hs[0]));
    return 6;
  }
  if (mpe_estimate_batch_multi(hs, n_dev, frames, n, rows, cols, cols, fb, markers, nm, K, D, 5, &p, many) != MPE_OK) {
    fprintf(stderr, "multi failed\n");
    return 7;
  }
  if (memcmp(one, many, sizeof(mpe_result) * n) != 0) return 8;
  int lo, hi, covered = 0;
  for (int d = 0; d < n_dev; ++d) {
    mpe_shard_bounds(n, d, n_dev, &lo, &hi);
    if (lo != covered) return 9;
    covered = hi;
  }
  if (covered != n) return 9;
  /* usage errors are loud */
  mpe_handle* dup[2] = {hs[0], hs[0]};
  if (mpe_estimate_batch_multi(dup, 2, frames, n, rows, cols, cols, fb, markers, nm, K, D, 5, &p, many) != MPE_ERR_ARG) return 10;
  if (mpe_estimate_batch_multi(hs, 0, frames, n, rows, cols, cols, fb, markers, nm, K, D, 5, &p, many) != MPE_ERR_ARG) return 11;
  int poses = 0;
  for (int i = 0; i < n; ++i) poses += one[i].status == MPE_FRAME_POSE;
  printf("multi ok: %d frames over %d shard(s) on %d GPU(s), %d poses\n", n, n_dev, n_gpu < n_dev ? n_gpu : n_dev, poses);
  for (int d = 0; d < n_dev; ++d) mpe_destroy(hs[d]);
  free(frames);
  free(one);
  free(many);
  return 0;
}
