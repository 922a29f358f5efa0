/* oracle/input_ref.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, host-memory restatement of the training-sample transform of the reference's Cityscapes loader,
 * dataset/datasets.py:173-210 (CSDataSet.__getitem__) after the PNG decode, behind the same C ABI as
 * skd_cs_transform() of include/skd.h.  Unlike the fused device kernel it MATERIALISES every intermediate exactly in
 * the reference's order: id -> trainId look-up (datasets.py:162-171), cv2.resize of image (INTER_LINEAR) and label
 * (INTER_NEAREST) (:157-160), float32 conversion and mean subtraction (:181-182), bottom / right padding (:183-194),
 * crop (:196-201), HWC -> CHW and mirror (:203-208).
 *
 * cv2.resize is a third-party dependency that is absent here (OpenCV, version not pinned by the reference: README.md
 * lists "cv2").  Its published 8-bit algorithm (modules/imgproc/src/resize.cpp) is restated:
 *   dsize       = (cvRound(W * fx), cvRound(H * fy))  -- computed by the caller with the same rounding (half to even)
 *   INTER_LINEAR: per destination column  fx = (float)((dx + 0.5) * (1 / f) - 0.5); sx = floor(fx); fx -= sx;
 *                 sx < 0 -> (0, fx = 0); sx >= W - 1 -> (W - 1, fx = 0); alpha = saturate_cast<short>({1 - fx, fx} * 2048);
 *                 horizontal pass in int: S = src[sx] * alpha0 + src[sx + 1] * alpha1 (src[sx] * 2048 at the right border);
 *                 rows: sy = floor(fy), rows clipped to [0, H - 1], beta = saturate_cast<short>({1 - fy, fy} * 2048);
 *                 dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
 *   INTER_NEAREST: src index = min(cvFloor(d * (1 / f)), size - 1)
 * PARITY UNPINNED against cv2 itself (not installable here); pinned against this restatement of the published source.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void *stream_t;

static int sat_short(float v) {
  long r = lrintf(v); /* round half to even (default rounding mode) */
  return r < -32768 ? -32768 : (r > 32767 ? 32767 : (int)r);
}

/* cv2.resize(src (H, W, cn) uint8, fx = fy = f, INTER_LINEAR) -> dst (dh, dw, cn) */
static int resize_linear_u8(const uint8_t *src, int H, int W, int cn, double f, uint8_t *dst, int dh, int dw) {
  const double inv = 1.0 / f;
  int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *a0 = (int *)malloc(sizeof(int) * (size_t)dw), *a1 = (int *)malloc(sizeof(int) * (size_t)dw);
  int *row0 = (int *)malloc(sizeof(int) * (size_t)dw * cn), *row1 = (int *)malloc(sizeof(int) * (size_t)dw * cn);
  if (!xofs || !a0 || !a1 || !row0 || !row1) return 0;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * inv - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= W - 1) { fx = 0.f; sx = W - 1; }
    xofs[dx] = sx;
    a0[dx] = sat_short((1.f - fx) * 2048.f);
    a1[dx] = sat_short(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * inv - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
    int r[2] = {sy, sy + 1};
    int *rows[2] = {row0, row1};
    for (int k = 0; k < 2; ++k) { /* horizontal pass of the two source rows, clipped to the image */
      const int rr = r[k] < 0 ? 0 : (r[k] > H - 1 ? H - 1 : r[k]);
      const uint8_t *S = src + (size_t)rr * W * cn;
      for (int dx = 0; dx < dw; ++dx)
        for (int c = 0; c < cn; ++c) {
          const int sx = xofs[dx];
          rows[k][dx * cn + c] = sx >= W - 1 ? (int)S[sx * cn + c] * 2048 : (int)S[sx * cn + c] * a0[dx] + (int)S[(sx + 1) * cn + c] * a1[dx];
        }
    }
    for (int q = 0; q < dw * cn; ++q)
      dst[(size_t)dy * dw * cn + q] = (uint8_t)((((b0 * (row0[q] >> 4)) >> 16) + ((b1 * (row1[q] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(a0); free(a1); free(row0); free(row1);
  return 1;
}

static void resize_nearest_u8(const uint8_t *src, int H, int W, double f, uint8_t *dst, int dh, int dw) {
  const double inv = 1.0 / f;
  for (int dy = 0; dy < dh; ++dy) {
    int sy = (int)floor(dy * inv);
    if (sy > H - 1) sy = H - 1;
    for (int dx = 0; dx < dw; ++dx) {
      int sx = (int)floor(dx * inv);
      if (sx > W - 1) sx = W - 1;
      dst[(size_t)dy * dw + dx] = src[(size_t)sy * W + sx];
    }
  }
}

int skd_cs_transform(int B, int H0, int W0, const uint8_t *images, const uint8_t *labels, const uint8_t *lut,
                     const double *scale, const int *dst_h, const int *dst_w, const int *h_off, const int *w_off,
                     const int *flip, int crop_h, int crop_w, const float *mean, int ignore_label, float *out_image,
                     int channels_last, int64_t *out_label, stream_t st) {
  (void)st;
  if (B <= 0 || H0 <= 0 || W0 <= 0 || crop_h <= 0 || crop_w <= 0) return 0;
  if (!images || !scale || !dst_h || !dst_w || !h_off || !w_off || !flip || !mean || !out_image) return 0;
  if ((labels != NULL) != (out_label != NULL) || (labels && !lut)) return 0;
  for (int b = 0; b < B; ++b) {
    const int dh = dst_h[b], dw = dst_w[b];
    const uint8_t *img = images + (size_t)b * H0 * W0 * 3;
    /* 1. id -> trainId on the full-size label */
    uint8_t *lab = NULL, *rlab = NULL;
    if (labels) {
      lab = (uint8_t *)malloc((size_t)H0 * W0);
      rlab = (uint8_t *)malloc((size_t)dh * dw);
      if (!lab || !rlab) return 0;
      for (size_t q = 0; q < (size_t)H0 * W0; ++q) lab[q] = lut[labels[(size_t)b * H0 * W0 + q]];
    }
    /* 2. resize (uint8 in, uint8 out) */
    uint8_t *rimg = (uint8_t *)malloc((size_t)dh * dw * 3);
    if (!rimg || !resize_linear_u8(img, H0, W0, 3, scale[b], rimg, dh, dw)) return 0;
    if (labels) resize_nearest_u8(lab, H0, W0, scale[b], rlab, dh, dw);
    /* 3. float32 - mean; 4. pad to >= crop (image 0.0, label ignore) */
    const int ph = dh < crop_h ? crop_h : dh, pw = dw < crop_w ? crop_w : dw;
    float *pimg = (float *)calloc((size_t)ph * pw * 3, sizeof(float));
    uint8_t *plab = labels ? (uint8_t *)malloc((size_t)ph * pw) : NULL;
    if (!pimg || (labels && !plab)) return 0;
    if (plab) memset(plab, ignore_label, (size_t)ph * pw);
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x) {
        for (int c = 0; c < 3; ++c) pimg[((size_t)y * pw + x) * 3 + c] = (float)rimg[((size_t)y * dw + x) * 3 + c] - mean[c];
        if (plab) plab[(size_t)y * pw + x] = rlab[(size_t)y * dw + x];
      }
    /* 5. crop, 6. HWC -> CHW, 7. mirror */
    if (h_off[b] < 0 || w_off[b] < 0 || h_off[b] + crop_h > ph || w_off[b] + crop_w > pw) return 0;
    for (int y = 0; y < crop_h; ++y)
      for (int x = 0; x < crop_w; ++x) {
        const int sx = flip[b] < 0 ? crop_w - 1 - x : x;
        const size_t s = (size_t)(y + h_off[b]) * pw + (sx + w_off[b]);
        for (int c = 0; c < 3; ++c) {
          const size_t o = channels_last ? (((size_t)b * crop_h + y) * crop_w + x) * 3 + c
                                         : (((size_t)b * 3 + c) * crop_h + y) * crop_w + x;
          out_image[o] = pimg[s * 3 + c];
        }
        if (out_label) out_label[((size_t)b * crop_h + y) * crop_w + x] = (int64_t)plab[s];
      }
    free(lab); free(rlab); free(rimg); free(pimg); free(plab);
  }
  return 1;
}
