/*
 * batch_intra.c -- plain C host program over the C ABI (include/kvz_hip_batch.h), the way a kvazaar-side caller would drive the
 * batched all-intra pass: read planar 4:2:0 frames, run search + reconstruction of every CTU on the GPU, deblock, print the
 * per-frame picture-hash checksums (nal.c:73-86) and the summed RD cost, optionally write the reconstruction.
 *
 *   gcc -O2 -Iinclude examples/batch_intra.c -Lkvazaar_amd/lib -lkvz_hip -Wl,-rpath,$PWD/kvazaar_amd/lib -o batch_intra
 *   ./batch_intra in.yuv 1920 1080 [qp=22] [out_rec.yuv]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "kvz_hip.h"
#include "kvz_hip_batch.h"

/* kvz_fast_coeff_get_weights(state) of kvazaar's default table at QP 22 (fast_coeff_cost.h default_fast_coeff_cost_wts); a real
 * integration passes the encoder's own value for its QP. */
#define COEFF_WEIGHTS_QP22 0x065403F0052C0004ull

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s in.yuv width height [qp] [out_rec.yuv]\n", argv[0]); return 2; }
  const int w = atoi(argv[2]), h = atoi(argv[3]), qp = argc > 4 ? atoi(argv[4]) : 22;
  const size_t ys = (size_t)w * h, cs = ys / 4, fs = ys + 2 * cs;
  if (w <= 0 || h <= 0 || (w & 7) || (h & 7)) { fprintf(stderr, "width and height must be positive multiples of 8\n"); return 2; }
  if (kvz_hip_device_count() < 1) { fprintf(stderr, "no usable gfx950 device\n"); return 1; }

  FILE *in = fopen(argv[1], "rb");
  if (!in) { perror(argv[1]); return 1; }
  fseek(in, 0, SEEK_END);
  const int n_frames = (int)((size_t)ftell(in) / fs);
  fseek(in, 0, SEEK_SET);
  if (n_frames < 1) { fprintf(stderr, "%s holds no complete %dx%d frame\n", argv[1], w, h); return 1; }

  kvz_hip_intra_cost_model model;
  kvz_hip_intra_cost_model_init(qp, COEFF_WEIGHTS_QP22, &model);
  kvz_hip_batch *b = kvz_hip_batch_create(w, h, n_frames);
  uint8_t *frame = malloc(fs);
  for (int i = 0; i < n_frames; i++) {
    if (fread(frame, 1, fs, in) != fs) { fprintf(stderr, "short read\n"); return 1; }
    kvz_hip_batch_upload(b, i, frame, frame + ys, frame + ys + cs);
  }
  fclose(in);

  kvz_hip_intra_frames(b, &model);          /* asynchronous: one persistent launch for the whole batch */
  kvz_hip_batch_deblock(b, qp, 0, 0);       /* same stream: runs behind the CTU pass */
  kvz_hip_batch_sync(b);
  printf("%d frame(s) %dx%d, qp %d: CTU pass %.3f ms on the device\n", n_frames, w, h, qp, kvz_hip_batch_last_kernel_ms(b));

  const int ctus = kvz_hip_batch_ctus_per_frame(b);
  int16_t *coeff = malloc((size_t)ctus * KVZ_HIP_CTU_COEFFS * sizeof(int16_t));
  uint8_t *depth = malloc((size_t)(w / 8) * (h / 8)), *mode = malloc((size_t)(w / 8) * (h / 8));
  double *cost = malloc((size_t)ctus * sizeof(double));
  FILE *out = argc > 5 ? fopen(argv[5], "wb") : NULL;
  for (int i = 0; i < n_frames; i++) {
    kvz_hip_batch_download(b, i, frame, frame + ys, frame + ys + cs, coeff, depth, mode, cost);
    double total = 0;
    for (int c = 0; c < ctus; c++) total += cost[c];
    printf("frame %d: checksum Y %08x U %08x V %08x  rd cost %.1f\n", i, kvz_hip_plane_checksum(frame, h, w, w),
           kvz_hip_plane_checksum(frame + ys, h / 2, w / 2, w / 2), kvz_hip_plane_checksum(frame + ys + cs, h / 2, w / 2, w / 2), total);
    if (out) fwrite(frame, 1, fs, out);
  }
  if (out) fclose(out);
  kvz_hip_batch_destroy(b);
  free(frame); free(coeff); free(depth); free(mode); free(cost);
  return 0;
}
