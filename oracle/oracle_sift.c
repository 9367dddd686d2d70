/* oracle/oracle_sift.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the SIFT leg of the hot path, sparse branch of
 * SIFT_FeatureFactory::generateFeatures (src/SIFT_FeatureFactory.cu:17-31,55-169) with the constants that
 * function hard-codes (:56-64) and the quirks listed in SURVEY.md section 7.
 *
 * Floating-point policy: plain single-precision expressions in the reference's operand order, no implicit
 * contraction (-ffp-contract=off).  The three accumulation loops where nvcc's default -fmad would fuse
 * `sum += a*b` (separable convolution, orientation histogram, nothing else) use fmaf() explicitly.
 */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_math.h"

#define NUM_OCT 4
#define NUM_GAUSS 6
#define NUM_DOG 5

typedef struct {
  uint32_t w, h;
  float pixelWidth;
  float sigma[NUM_GAUSS];       /* label of gaussian level b (also label of dog level b, FeatureFactory.cu:423) */
  float* gauss[NUM_GAUSS];      /* normalised gaussian levels (after Octave::normalize) */
  float gmin[NUM_GAUSS], gmax[NUM_GAUSS];
  float* dog[NUM_DOG];          /* raw DoG (before findKeyPoints' second normalisation) */
  float dmin[NUM_DOG], dmax[NUM_DOG];
  float* dogn[NUM_DOG];         /* DoG after the second normalisation (FeatureFactory.cu:472) */
  o_sskeypoint* extrema;        /* NULL == reference's nullptr */
  int n;
  int idx[NUM_DOG];             /* extremaBlurIndices */
} octave_t;

struct oracle_sift {
  uint32_t W, H;
  octave_t oct[NUM_OCT];
};

/* src/Image.cu:1248-1252 */
static inline int sym_coord(int i, unsigned int l) {
  int ll = 2 * (int)l;
  i = (i + ll) % ll;
  return ((unsigned int)i > l - 1) ? ll - 1 - i : i;
}

/* src/Image.cu:1393-1414 upsampleImage(float) */
static float* upsample2x(const float* in, uint32_t w, uint32_t h) {
  float* out = (float*)malloc(sizeof(float) * 4 * (size_t)w * h);
#pragma omp parallel for
  for (uint32_t j = 0; j < h * 2; ++j) {
    for (uint32_t i = 0; i < w * 2; ++i) {
      float x = i * 0.5f;
      float y = j * 0.5f;
      int xm = sym_coord((int)x, w);
      int xp = sym_coord((int)x + 1, w);
      int ym = sym_coord((int)y, h);
      int yp = sym_coord((int)y + 1, h);
      float dx = x - floorf(x), dy = y - floorf(y);
      float sumPix = dx * dy * in[(size_t)yp * w + xp];
      sumPix += (1.0f - dx) * dy * in[(size_t)yp * w + xm];
      sumPix += dx * (1 - dy) * in[(size_t)ym * w + xp];
      sumPix += (1 - dx) * (1 - dy) * in[(size_t)ym * w + xm];
      out[(size_t)j * (w * 2) + i] = sumPix;
    }
  }
  return out;
}

/* src/Image.cu:1380-1392 binImage(float) */
static float* bin2x(const float* in, uint32_t w, uint32_t h) {
  uint32_t ow = w / 2, oh = h / 2;
  float* out = (float*)malloc(sizeof(float) * (size_t)ow * oh);
#pragma omp parallel for
  for (uint32_t y = 0; y < oh; ++y) {
    for (uint32_t x = 0; x < ow; ++x) {
      float sumPix = in[(size_t)y * 2 * w + x * 2] + in[(size_t)(y * 2 + 1) * w + x * 2] +
                     in[(size_t)y * 2 * w + (x * 2 + 1)] + in[(size_t)(y * 2 + 1) * w + (x * 2 + 1)];
      out[(size_t)y * ow + x] = sumPix / 4.0f;
    }
  }
  return out;
}

/* Gaussian taps of Blur::Blur (src/FeatureFactory.cu:15-18,29-33).  Returns odd tap count, fills weights. */
int oracle_gauss_kernel(float sigma, float pixelWidth, float* weights /* >= 129 */) {
  int ksize = (int)ceilf((float)8 * sigma / pixelWidth);
  if (ksize % 2 == 0) ksize++;
  int i = 0;
  for (int x = -ksize / 2; x <= ksize / 2; ++x, ++i) {
    weights[i] = expf(-((x * x) / 2.0f / sigma / sigma)) / sqrtf((float)(2.0f * O_PI)) / sigma;
  }
  return ksize;
}

/* src/Image.cu:1197-1239 convolveSeparable + :1526-1546 convolveImage1D_symmetric (horizontal then vertical).
 * `sum += p*k` is written fmaf(p,k,sum): nvcc contracts it (-fmad=true default). */
static float* conv_separable(const float* in, uint32_t w, uint32_t h, int ksize, const float* kernel) {
  float* half = (float*)malloc(sizeof(float) * (size_t)w * h);
  float* out = (float*)malloc(sizeof(float) * (size_t)w * h);
  int r = ksize / 2;
#pragma omp parallel for
  for (uint32_t y = 0; y < h; ++y) {
    for (uint32_t x = 0; x < w; ++x) {
      float sum = 0.0f;
      for (int kx = -r; kx <= r; ++kx) {
        int sx = sym_coord((int)x + kx, w);
        sum = fmaf(in[(size_t)y * w + sx], kernel[kx + r], sum);
      }
      half[(size_t)y * w + x] = sum;
    }
  }
#pragma omp parallel for
  for (uint32_t y = 0; y < h; ++y) {
    for (uint32_t x = 0; x < w; ++x) {
      float sum = 0.0f;
      for (int ky = -r; ky <= r; ++ky) {
        int sy = sym_coord((int)y + ky, h);
        sum = fmaf(half[(size_t)sy * w + x], kernel[ky + r], sum);
      }
      out[(size_t)y * w + x] = sum;
    }
  }
  free(half);
  return out;
}

/* src/Image.cu:631-649 normalizeImage + :1560-1565 normalize */
static void minmax(const float* p, size_t n, float* mn, float* mx) {
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (size_t i = 0; i < n; ++i) {
    if (lo > p[i]) lo = p[i];
    if (hi < p[i]) hi = p[i];
  }
  *mn = lo;
  *mx = hi;
}
static void normalize_level(float* p, size_t n, float mn, float mx) {
#pragma omp parallel for
  for (size_t i = 0; i < n; ++i) p[i] = (p[i] - mn) / (mx - mn);
}

/* S3: src/Image.cu:572-598 addBufferBorder(float).  The new buffer comes from Unity's nullptr constructor with state gpu,
 * which fills it from a value-initialised array (include/Unity.cuh:763-790): the border is 0.0f. */
static float* add_border(const float* in, uint32_t w, uint32_t h, int bx, int by) {
  uint32_t nw = w + 2 * (uint32_t)bx, nh = h + 2 * (uint32_t)by;
  float* out = (float*)calloc((size_t)nw * nh, sizeof(float));
  for (uint32_t y = 0; y < h; ++y) memcpy(out + (size_t)(y + by) * nw + bx, in + (size_t)y * w, sizeof(float) * w);
  return out;
}
/* S3: src/Image.cu:966-995 makeBinnable(float): pad so that both sides are multiples of 2^plannedDepth.  Takes
 * ownership of `px`, updates the size, returns the (possibly new) image. */
static float* make_binnable(float* px, uint32_t* w, uint32_t* h, int plannedDepth) {
  int numResize = 1 << plannedDepth; /* (int)pow(2, plannedDepth) */
  int off[2] = {(int)(*w % (uint32_t)numResize), (int)(*h % (uint32_t)numResize)};
  if (!off[0] && !off[1]) return px;
  int mustSizeUp = (*w % 2) || (*h % 2); /* never true from ScaleSpace (even sizes reach here), restated for fidelity */
  if (mustSizeUp) {
    float* up = upsample2x(px, *w, *h);
    free(px);
    px = up;
    *w *= 2;
    *h *= 2;
    numResize *= 2;
    off[0] = (int)(*w % (uint32_t)numResize);
    off[1] = (int)(*h % (uint32_t)numResize);
  }
  int bx = off[0] ? (numResize - (int)(*w % (uint32_t)numResize)) / 2 : 0;
  int by = off[1] ? (numResize - (int)(*h % (uint32_t)numResize)) / 2 : 0;
  float* padded = add_border(px, *w, *h, bx, by);
  free(px);
  px = padded;
  *w += 2 * (uint32_t)bx;
  *h += 2 * (uint32_t)by;
  if (mustSizeUp) {
    float* b = bin2x(px, *w, *h);
    free(px);
    px = b;
    *w /= 2;
    *h /= 2;
  }
  return px;
}

oracle_sift* oracle_sift_create(const uint8_t* pixels, uint32_t width, uint32_t height) {
  /* ScaleSpace::ScaleSpace "too small" test (src/FeatureFactory.cu:341-345): numResize = 2^(startingOctave + depth.x) */
  if (width / 8 == 0 || height / 8 == 0) return NULL;
  oracle_sift* s = (oracle_sift*)calloc(1, sizeof *s);
  s->W = width;
  s->H = height;
  /* S1: src/Image.cu:1554-1559 */
  size_t n = (size_t)width * height;
  float* flt = (float*)malloc(sizeof(float) * n);
  for (size_t i = 0; i < n; ++i) flt[i] = (float)pixels[i];
  /* S3 + S2 (src/FeatureFactory.cu:364-376), startingOctave = -1, depth.x = 4: even sizes are padded to multiples of
   * 2^3 before the one 2x upsample; sizes with an odd side are upsampled first and then padded to multiples of 2^5
   * (makeBinnable(imageSize, pixels, depth.x - i) with i = -1). */
  uint32_t w = width, h = height;
  int canBinEarly = (w % 2 == 0) && (h % 2 == 0);
  if (canBinEarly) flt = make_binnable(flt, &w, &h, -1 + NUM_OCT);
  float* cur = upsample2x(flt, w, h);
  free(flt);
  w *= 2;
  h *= 2;
  if (!canBinEarly) cur = make_binnable(cur, &w, &h, NUM_OCT + 1);
  float pixelWidth = 1.0f;
  pixelWidth /= 2.0f;
  /* FeatureFactory.cu:383-387 with SIFT_FeatureFactory.cu:63-64 constants */
  float sigmas[NUM_GAUSS];
  float mulY = sqrtf(2.0f), mulX = 2;
  sigmas[0] = sqrtf(2.0f) / 2.0f;
  for (int i = 1; i < NUM_GAUSS; ++i) sigmas[i] = sigmas[i - 1] * mulY;
  float weights[257];
  for (int o = 0; o < NUM_OCT; ++o) {
    octave_t* oc = &s->oct[o];
    oc->w = w;
    oc->h = h;
    oc->pixelWidth = pixelWidth;
    /* Octave::Octave (FeatureFactory.cu:63-82): levels 0..3 cumulative on `pixels`, 4..5 on a copy */
    const float* src = cur;
    for (int b = 0; b < NUM_GAUSS; ++b) {
      oc->sigma[b] = sigmas[b];
      int ks = oracle_gauss_kernel(sigmas[b], pixelWidth, weights);
      oc->gauss[b] = conv_separable(src, w, h, ks, weights);
      src = oc->gauss[b];
    }
    free(cur);
    cur = NULL;
    if (o + 1 < NUM_OCT) {
      /* FeatureFactory.cu:392-399: bin the un-normalised level 3 (pixels aliases blurs[3]) */
      cur = bin2x(oc->gauss[3], w, h);
    }
    /* FeatureFactory.cu:400 octave.normalize() */
    for (int b = 0; b < NUM_GAUSS; ++b) {
      minmax(oc->gauss[b], (size_t)w * h, &oc->gmin[b], &oc->gmax[b]);
      normalize_level(oc->gauss[b], (size_t)w * h, oc->gmin[b], oc->gmax[b]);
    }
    /* convertToDOG (FeatureFactory.cu:404-440, :842-845) */
    for (int b = 0; b < NUM_DOG; ++b) {
      oc->dog[b] = (float*)malloc(sizeof(float) * (size_t)w * h);
      const float* up = oc->gauss[b + 1];
      const float* lo = oc->gauss[b];
      float* d = oc->dog[b];
#pragma omp parallel for
      for (size_t i = 0; i < (size_t)w * h; ++i) d[i] = up[i] - lo[i];
      /* second normalisation (findKeyPoints, FeatureFactory.cu:472) precomputed here */
      minmax(d, (size_t)w * h, &oc->dmin[b], &oc->dmax[b]);
      oc->dogn[b] = (float*)malloc(sizeof(float) * (size_t)w * h);
      float* dn = oc->dogn[b];
      float mn = oc->dmin[b], mx = oc->dmax[b];
#pragma omp parallel for
      for (size_t i = 0; i < (size_t)w * h; ++i) dn[i] = (d[i] - mn) / (mx - mn);
    }
    if (o + 1 < NUM_OCT) {
      w /= 2;
      h /= 2;
      pixelWidth *= 2.0f;
      for (int b = 0; b < NUM_GAUSS; ++b) sigmas[b] *= mulX;
    }
  }
  return s;
}

void oracle_sift_destroy(oracle_sift* s) {
  if (!s) return;
  for (int o = 0; o < NUM_OCT; ++o) {
    for (int b = 0; b < NUM_GAUSS; ++b) free(s->oct[o].gauss[b]);
    for (int b = 0; b < NUM_DOG; ++b) { free(s->oct[o].dog[b]); free(s->oct[o].dogn[b]); }
    free(s->oct[o].extrema);
  }
  free(s);
}

int oracle_sift_level(const oracle_sift* s, int kind, int octave, int blur, float* out, uint32_t* w, uint32_t* h) {
  const octave_t* oc = &s->oct[octave];
  const float* src = kind == 0 ? oc->gauss[blur] : kind == 1 ? oc->dog[blur] : oc->dogn[blur];
  if (w) *w = oc->w;
  if (h) *h = oc->h;
  if (out) memcpy(out, src, sizeof(float) * (size_t)oc->w * oc->h);
  return 0;
}
void oracle_sift_minmax(const oracle_sift* s, int kind, int octave, int blur, float* mn, float* mx) {
  const octave_t* oc = &s->oct[octave];
  if (kind == 0) { *mn = oc->gmin[blur]; *mx = oc->gmax[blur]; }
  else { *mn = oc->dmin[blur]; *mx = oc->dmax[blur]; }
}
void oracle_sift_octave_info(const oracle_sift* s, int octave, uint32_t* w, uint32_t* h, float* pixelWidth,
                             float* sigmas6) {
  const octave_t* oc = &s->oct[octave];
  *w = oc->w; *h = oc->h; *pixelWidth = oc->pixelWidth;
  memcpy(sigmas6, oc->sigma, sizeof oc->sigma);
}

/* ---------------------------------------------------------------------------------------------------- */
/* Octave::discardExtrema (src/FeatureFactory.cu:161-215): per-blur stable compaction, indices rebuilt */
static void discard_extrema(octave_t* oc) {
  if (!oc->extrema) return;
  int cnt[NUM_DOG];
  for (int i = 0; i < NUM_DOG; ++i) cnt[i] = (i < NUM_DOG - 1) ? oc->idx[i + 1] - oc->idx[i] : oc->n - oc->idx[i];
  o_sskeypoint* out = (o_sskeypoint*)malloc(sizeof(o_sskeypoint) * (oc->n ? oc->n : 1));
  int total = 0;
  for (int i = 0; i < NUM_DOG; ++i) {
    int start = oc->idx[i];
    oc->idx[i] = total;
    for (int k = 0; k < cnt[i]; ++k)
      if (!oc->extrema[start + k].discard) out[total++] = oc->extrema[start + k];
  }
  free(oc->extrema);
  if (total) { oc->extrema = out; oc->n = total; }
  else { free(out); oc->extrema = NULL; oc->n = 0; }
}

/* Octave::searchForExtrema (src/FeatureFactory.cu:86-159) + findExtrema/fillExtrema (:847-890) */
static void search_extrema(octave_t* oc, int octaveId) {
  uint32_t w = oc->w, h = oc->h;
  int cap = 1 << 16, n = 0;
  o_sskeypoint* list = (o_sskeypoint*)malloc(sizeof(o_sskeypoint) * cap);
  oc->idx[0] = 0;
  for (int b = 1; b < NUM_DOG - 1; ++b) {
    const float* lo = oc->dog[b - 1];
    const float* mid = oc->dog[b];
    const float* up = oc->dog[b + 1];
    oc->idx[b] = n;
    for (uint32_t y = 1; y + 1 < h; ++y) {
      for (uint32_t x = 1; x + 1 < w; ++x) {
        float value = mid[(size_t)y * w + x];
        float mx = -FLT_MAX, mn = FLT_MAX;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            size_t a = (size_t)(y + dy) * w + (x + dx);
            float v0 = lo[a], v1 = mid[a], v2 = up[a];
            if (v0 > mx) mx = v0;
            if (v0 < mn) mn = v0;
            if (v1 > mx) mx = v1;
            if (v1 < mn) mn = v1;
            if (v2 > mx) mx = v2;
            if (v2 < mn) mn = v2;
          }
        if (mx == value || mn == value) { /* non-strict: ties count (:873) */
          if (n == cap) { cap *= 2; list = (o_sskeypoint*)realloc(list, sizeof(o_sskeypoint) * cap); }
          o_sskeypoint kp;
          memset(&kp, 0, sizeof kp);
          kp.octave = octaveId;
          kp.blur = b;
          kp.loc.x = (float)x;
          kp.loc.y = (float)y;
          kp.intensity = value;
          kp.sigma = oc->sigma[b];
          kp.theta = -1.0f;
          kp.discard = 0;
          list[n++] = kp;
        }
      }
    }
  }
  if (n) {
    oc->extrema = list;
    oc->n = n;
    oc->idx[NUM_DOG - 1] = n;
  } else {
    free(list);
    oc->extrema = NULL;
    oc->n = 0;
  }
}

/* flagNoise (src/FeatureFactory.cu:968-973) + removeNoise (:266-278) */
static void remove_noise(octave_t* oc, float thr) {
  if (!oc->extrema) return;
  for (int i = 0; i < oc->n; ++i) oc->extrema[i].discard = fabsf(oc->extrema[i].intensity) < thr;
  discard_extrema(oc);
}

/* refineLocation (src/FeatureFactory.cu:892-967) on the twice-normalised DoG levels */
static void refine_one(const octave_t* oc, o_sskeypoint* pkp) {
  o_sskeypoint kp = *pkp;
  int W = (int)oc->w, H = (int)oc->h;
  unsigned int numBlurs = NUM_DOG;
  float sigmaMin = oc->sigma[0];
  float mult = oc->sigma[1] / oc->sigma[0];
  int lx = (int)roundf(kp.loc.x), ly = (int)roundf(kp.loc.y);
  float hess[3][3], hinv[3][3] = {{0}}, grad[3], temp[3], offset[3] = {0, 0, 0};
  const float* pl = oc->dogn[kp.blur - 1];
  const float* pm = oc->dogn[kp.blur];
  const float* pu = oc->dogn[kp.blur + 1];
  for (int attempt = 0; attempt < 5; ++attempt) {
#define PX(p, yy, xx) (p)[(size_t)(yy) * W + (xx)]
    grad[0] = PX(pm, ly, lx + 1) - PX(pm, ly, lx - 1);
    grad[1] = PX(pm, ly + 1, lx) - PX(pm, ly - 1, lx);
    grad[2] = PX(pu, ly, lx) - PX(pl, ly, lx);
    hess[0][0] = grad[0] - 2 * PX(pm, ly, lx);
    hess[0][1] = (PX(pm, ly + 1, lx + 1) - PX(pm, ly - 1, lx + 1) - PX(pm, ly + 1, lx - 1) + PX(pm, ly - 1, lx - 1)) / 4.0f;
    hess[0][2] = (PX(pu, ly, lx + 1) - PX(pl, ly, lx + 1) - PX(pu, ly, lx - 1) + PX(pl, ly, lx - 1)) / 4.0f;
    hess[1][0] = hess[0][1];
    hess[1][1] = grad[1] - 2 * PX(pm, ly, lx);
    hess[1][2] = (PX(pu, ly + 1, lx) - PX(pl, ly + 1, lx) - PX(pu, ly - 1, lx) + PX(pl, ly - 1, lx)) / 4.0f;
    hess[2][0] = hess[0][2];
    hess[2][1] = hess[1][2];
    hess[2][2] = grad[2] - 2 * PX(pm, ly, lx);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) hess[r][c] *= -1.0f;
    inverse3(hess, hinv); /* on det == 0 hinv keeps its previous contents, like the reference */
    for (int r = 0; r < 3; ++r) { /* matrix_util.cu:63-71 multiply(A[3][3], B[3]) */
      float val = 0;
      for (int c = 0; c < 3; ++c) val = fmaf(hinv[r][c], grad[c], val); /* nvcc's fused chain (oracle_math.h) */
      offset[r] = val;
    }
    for (int c = 0; c < 3; ++c) { /* matrix_util.cu:73-81 multiply(A[3], B[3][3]) */
      float val = 0;
      for (int r = 0; r < 3; ++r) val = fmaf(hess[r][c], grad[r], val);
      temp[c] = val;
    }
    if (fabsf(offset[0]) <= 0.5f && fabsf(offset[1]) <= 0.5f && fabsf(offset[2]) <= 0.5f) {
      kp.loc.x = (float)lx + offset[0];
      kp.loc.y = (float)ly + offset[1];
      lx = (int)roundf(kp.loc.x);
      ly = (int)roundf(kp.loc.y);
      kp.discard = (lx <= 0 || ly <= 0 || lx >= W - 1 || ly >= H - 1);
      if (kp.discard) break;
      kp.intensity = PX(pm, ly, lx) - (0.5f * ((temp[0] * grad[0]) + (temp[1] * grad[1]) + (temp[2] * grad[2])));
      kp.sigma = sigmaMin * sv_powf(mult, ((float)kp.blur + offset[2]));
      if (fabsf(offset[2]) > 0.5) kp.blur += (offset[2] > 0) ? 1 : -1;
      break;
    } else if (attempt == 4) {
      kp.discard = 1;
      break;
    } else {
      if (fabsf(offset[0]) > 0.5) lx += (offset[0] > 0) ? 1 : -1;
      if (fabsf(offset[1]) > 0.5) ly += (offset[1] > 0) ? 1 : -1;
      kp.loc.x = (float)lx;
      kp.loc.y = (float)ly;
      if (fabsf(offset[2]) > 0.5) kp.blur += (offset[2] > 0) ? 1 : -1;
      if (kp.blur >= (int)numBlurs - 1 || kp.blur <= 0 || lx <= 0 || ly <= 0 || lx >= W - 1 || ly >= H - 1) {
        kp.discard = 1;
        break;
      }
      pl = oc->dogn[kp.blur - 1];
      pm = oc->dogn[kp.blur];
      pu = oc->dogn[kp.blur + 1];
    }
#undef PX
  }
  *pkp = kp;
}

/* Octave::refineExtremaLocation (src/FeatureFactory.cu:217-265) */
static void refine_extrema(octave_t* oc) {
  if (!oc->extrema) return;
  for (int i = 0; i < oc->n; ++i) refine_one(oc, &oc->extrema[i]);
  discard_extrema(oc);
  if (!oc->extrema) return;
  /* thrust::stable_sort by blur (:250) */
  o_sskeypoint* tmp = (o_sskeypoint*)malloc(sizeof(o_sskeypoint) * oc->n);
  int k = 0;
  int lo = oc->extrema[0].blur, hi = lo;
  for (int i = 0; i < oc->n; ++i) { if (oc->extrema[i].blur < lo) lo = oc->extrema[i].blur; if (oc->extrema[i].blur > hi) hi = oc->extrema[i].blur; }
  for (int b = lo; b <= hi; ++b)
    for (int i = 0; i < oc->n; ++i)
      if (oc->extrema[i].blur == b) tmp[k++] = oc->extrema[i];
  free(oc->extrema);
  oc->extrema = tmp;
  /* host rescan of blur boundaries (:251-259); entries not reached keep discardExtrema's values */
  oc->idx[0] = 0;
  oc->idx[1] = 0;
  for (int i = 1, blur = 2; i < oc->n && blur < NUM_DOG - 1; ++i) {
    if (oc->extrema[i - 1].blur < oc->extrema[i].blur) oc->idx[blur++] = i;
  }
  oc->idx[NUM_DOG - 1] = oc->n;
}

/* flagEdges (src/FeatureFactory.cu:974-990) + removeEdges (:279-309) */
static void remove_edges(octave_t* oc, float thr) {
  if (!oc->extrema) return;
  int W = (int)oc->w;
  for (int i = 0; i < NUM_DOG; ++i) {
    int cnt = (i < NUM_DOG - 1) ? oc->idx[i + 1] - oc->idx[i] : oc->n - oc->idx[i];
    if (cnt == 0) continue;
    const float* px = oc->dogn[i];
    for (int k = 0; k < cnt; ++k) {
      o_sskeypoint* kp = &oc->extrema[oc->idx[i] + k];
      int lx = (int)roundf(kp->loc.x), ly = (int)roundf(kp->loc.y);
      float h00, h11, h01;
      h00 = -2.0f * px[(size_t)ly * W + lx];
      h11 = h00 + px[(size_t)(ly + 1) * W + lx] + px[(size_t)(ly - 1) * W + lx];
      h00 += px[(size_t)ly * W + lx + 1] + px[(size_t)ly * W + lx - 1];
      h01 = (px[(size_t)(ly + 1) * W + lx + 1] - px[(size_t)(ly - 1) * W + lx + 1] - px[(size_t)(ly + 1) * W + lx - 1] +
             px[(size_t)(ly - 1) * W + lx - 1]);
      float e = h00 + h11;                       /* trace, matrix_util.cu:217-219 */
      float det = (h00 * h11) - (h01 * h01);     /* determinant, matrix_util.cu:214-216 */
      kp->discard = (e * e / det) > thr;
    }
  }
  discard_extrema(oc);
}

/* checkKeyPoints (src/SIFT_FeatureFactory.cu:449-461) over every blur segment, then discardExtrema (:81-107) */
static void check_keypoints(octave_t* oc, float lambda) {
  if (!oc->extrema) return;
  for (int b = 0; b < NUM_DOG; ++b) {
    int cnt = (b + 1 == NUM_DOG) ? oc->n - oc->idx[b] : oc->idx[b + 1] - oc->idx[b];
    if (cnt == 0) continue;
    for (int k = 0; k < cnt; ++k) {
      o_sskeypoint* kp = &oc->extrema[oc->idx[b] + k];
      float ww = kp->sigma * lambda / oc->pixelWidth;
      if ((kp->loc.x - ww) < 0.0f || (kp->loc.y - ww) < 0.0f || (kp->loc.x + ww) >= oc->w - 1 ||
          (kp->loc.y + ww) >= oc->h - 1) {
        kp->discard = 1;
      }
    }
  }
  discard_extrema(oc);
}

/* calculatePixelGradients(float) (src/Image.cu:1583-1598) evaluated at one pixel of a level */
static inline o_float2 pixel_gradient(const float* px, int W, int H, int x, int y) {
  int xc0 = x + 1, xc1 = x - 1, yc0 = y + 1, yc1 = y - 1;
  if (xc1 == -1) { xc0 += 1; xc1 += 1; }
  else if (xc0 == W) { xc0 -= 1; xc1 -= 1; }
  if (yc1 == -1) { yc0 += 1; yc1 += 1; }
  else if (yc0 == H) { yc0 -= 1; yc1 -= 1; }
  o_float2 g;
  g.x = px[(size_t)y * W + xc0] - px[(size_t)y * W + xc1];
  g.y = px[(size_t)yc0 * W + x] - px[(size_t)yc1 * W + x];
  return g;
}

/* computeThetas(SSKeyPoint) (src/FeatureFactory.cu:1004-1112).  Writes up to maxOrientations thetas (-FLT_MAX = none). */
static void compute_thetas(const octave_t* oc, const float* level, const o_sskeypoint* kp, float lambda,
                           unsigned int maxOrientations, float orientationThreshold, float* thetas, int* valid) {
  const float pi = O_PI_F;
  int W = (int)oc->w, H = (int)oc->h;
  int regNumOrient = (int)maxOrientations;
  float kx = kp->loc.x, ky = kp->loc.y;
  float windowWidth = ceilf(kp->sigma * 3.0f * lambda / oc->pixelWidth);
  float minx = kx - windowWidth, miny = ky - windowWidth, maxx = kx + windowWidth, maxy = ky + windowWidth;
  for (int i = 0; i < regNumOrient; ++i) { thetas[i] = -FLT_MAX; valid[i] = 0; }
  if (minx < 0.0f || miny < 0.0f || maxx >= oc->w - 1 || maxy >= oc->h - 1) return;
  float hist[36];
  for (int i = 0; i < 36; ++i) hist[i] = 0.0f;
  float maxHist = 0.0f;
  float weight = 2.0f * lambda * lambda * kp->sigma * kp->sigma;
  float rad10 = pi / 18.0f;
  for (float y = miny; y <= maxy; y += 1.0f) {
    for (float x = minx; x <= maxx; x += 1.0f) {
      o_float2 g = pixel_gradient(level, W, H, (int)llroundf(x), (int)llroundf(y));
      float tx = x - kx, ty = y - ky;
      float angle = fmodf(sv_atan2f(g.y, g.x) + (2.0f * pi), 2.0f * pi);
      int bin = (int)floorf(angle / rad10);
      float mag = sqrtf((g.x * g.x) + (g.y * g.y));
      float wgt = sv_expf(-((tx * tx) + (ty * ty)) / weight);
      if (bin >= 0 && bin < 36) hist[bin] = fmaf(mag, wgt, hist[bin]); /* bin 36 would be an OOB write upstream */
    }
  }
  for (int i = 0; i < 36; ++i)
    if (hist[i] > maxHist) maxHist = hist[i];
  maxHist *= orientationThreshold;
  o_float2 best[8];
  for (int i = 0; i < regNumOrient; ++i) { best[i].x = 0.0f; best[i].y = 0.0f; }
  o_float2 t = {0.0f, 0.0f}, t2;
  for (int b = 0; b < 36; ++b) {
    if (hist[b] < maxHist || (b > 0 && hist[b] < hist[b - 1]) || (b < 35 && hist[b] < hist[b + 1]) ||
        (b == 0 && hist[b] < hist[35]) || (b == 35 && hist[b] < hist[0]) || (hist[b] < best[regNumOrient - 1].x)) {
      continue;
    }
    t.x = hist[b];
    if (b == 0) t.y = (hist[35] - hist[1]) / (hist[35] - (2.0f * hist[0]) + hist[1]);
    else if (b == 35) t.y = (hist[34] - hist[0]) / (hist[34] - (2.0f * hist[35]) + hist[0]);
    else t.y = (hist[b - 1] - hist[b + 1]) / (hist[b - 1] - (2.0f * hist[b]) + hist[b + 1]);
    t.y *= (pi / 36.0f);
    t.y += (b * rad10);
    t.y = fmodf(t.y + (2.0f * pi), 2.0f * pi);
    for (int i = 0; i < regNumOrient; ++i) {
      if (t.x > best[i].x) {
        for (int ii = i; ii < regNumOrient; ++ii) {
          t2 = best[ii];
          best[ii] = t;
          t = t2;
        }
      }
    }
  }
  for (int i = 0; i < regNumOrient; ++i) {
    if (best[i].x == 0.0f) { valid[i] = 0; thetas[i] = -FLT_MAX; }
    else { valid[i] = 1; thetas[i] = best[i].y; }
  }
}

/* ScaleSpace::computeKeyPointOrientations (src/FeatureFactory.cu:540-632) for one octave */
static void compute_orientations(octave_t* oc, float orientationThreshold, unsigned int maxOrientations, float lambda) {
  if (!oc->extrema) return;
  if (maxOrientations > 8) maxOrientations = 8;
  int cap = oc->n * (int)maxOrientations + 1, total = 0;
  o_sskeypoint* out = (o_sskeypoint*)malloc(sizeof(o_sskeypoint) * cap);
  for (int b = 0; b < NUM_DOG; ++b) {
    int cnt = (b + 1 != NUM_DOG) ? oc->idx[b + 1] - oc->idx[b] : oc->n - oc->idx[b];
    int keyPointIndex = oc->idx[b];
    oc->idx[b] = total;
    if (cnt <= 0) continue;
    for (int k = 0; k < cnt; ++k) {
      float thetas[8];
      int valid[8];
      const o_sskeypoint* kp = &oc->extrema[keyPointIndex + k];
      compute_thetas(oc, oc->dogn[b], kp, lambda, maxOrientations, orientationThreshold, thetas, valid);
      for (unsigned int i = 0; i < maxOrientations; ++i) {
        if (!valid[i]) continue;
        out[total] = *kp;
        out[total].theta = thetas[i];
        ++total;
      }
    }
  }
  free(oc->extrema);
  if (total) { oc->extrema = out; oc->n = total; }
  else { free(out); oc->extrema = NULL; oc->n = 0; }
}

/* fillDescriptors(SSKeyPoint) (src/SIFT_FeatureFactory.cu:475-549) for one keypoint.
 *
 * Every vote `temp` is the reference's expression, operand for operand.  What the reference leaves UNDEFINED is the
 * order of its sums: the 4x4x8 bins and both norms are accumulated with shared-memory float atomicAdd (:521,:533,:539)
 * from 16 resp. 128 threads, so two runs of the reference itself differ in the last bits (its own two fixture runs
 * differ by one descriptor byte).  A restatement has to pick an order; two are offered:
 *   sum mode 0 (default): order-INDEPENDENT.  A vote enters its bin as the integer nearest to temp * 2^k (halves
 *     round up; k chosen per key point so that no bin can reach 2^31: k = 22 at w = 12, 20 at w = 29); the integer
 *     sum is exact, hence the same in any order -- the one definition a parallel implementation can reproduce bit
 *     for bit.  (Rounding matters: with truncation every bin sits low by half a unit per vote, enough to move
 *     descriptor bytes and to lose 2 of the 13 534 golden matches.)  The two norms are summed as a balanced binary
 *     tree over the bins in [nx][ny][k] order (pairs 64 apart first, then 32, 16, ... 1).
 *   sum mode 1: float sums, samples in raster order, norm in CUDA linear thread order (the round-1 oracle).
 * tests/test_oracle_golden.py pins BOTH against the reference's golden match sets (13 534 / 21 177 exactly). */
static int g_desc_sum_mode = 0;
void oracle_set_descriptor_sum_mode(int mode) { g_desc_sum_mode = mode; }

/* fixed-point exponent of sum mode 0: the largest k with sqrt(2) (w + 2)^2 2^k < 2^31 (a bin receives fewer than
 * (2 binWidth + 2)^2 = (w + 2)^2 votes, each at most sqrt(2): gradient of a [0,1]-normalised level) */
static int desc_vote_exponent(float windowWidth) {
  int boundExp;
  (void)frexpf(1.4143f * ((windowWidth + 2.0f) * (windowWidth + 2.0f)), &boundExp);
  return 31 - boundExp;
}

/* nearest integer, halves up (exact: q - floorf(q) is exact in float) */
static inline uint32_t vote_fixed(float q) {
  float f = floorf(q);
  return (uint32_t)f + ((q - f) >= 0.5f ? 1u : 0u);
}

/* sum of 128 squares as the balanced tree described above */
static float tree_norm(const float* v /* [128], [nx][ny][k] order */) {
  float p[64];
  for (int i = 0; i < 64; ++i) p[i] = (v[i] * v[i]) + (v[i + 64] * v[i + 64]);
  for (int o = 32; o > 0; o >>= 1)
    for (int i = 0; i < o; ++i) p[i] = p[i] + p[i + o];
  return p[0];
}

void oracle_fill_descriptor(const float* level, uint32_t W, uint32_t H, float pixelWidth, float lambda,
                            const o_sskeypoint* kp, o_feature* feat) {
  const float pi = O_PI_F;
  float bins[4][4][8];
  uint32_t ibins[4][4][8];
  memset(bins, 0, sizeof bins);
  memset(ibins, 0, sizeof ibins);
  const int exact = g_desc_sum_mode == 0;
  float kx = kp->loc.x, ky = kp->loc.y;
  float windowWidth = ceilf(kp->sigma * lambda / pixelWidth);
  float theta = kp->theta;
  float binWidth = windowWidth / 2.0f;
  float rad45 = pi / 4.0f;
  float c = sv_cosf(-theta), s = sv_sinf(-theta);
  const float voteScale = ldexpf(1.0f, desc_vote_exponent(windowWidth));
  for (float y = -windowWidth; y <= windowWidth; y += 1.0f) {
    for (float x = -windowWidth; x <= windowWidth; x += 1.0f) {
      float cx = (x * c) + (y * s), cy = (-x * s) + (y * c);
      if (fabsf(cx) > windowWidth || fabsf(cy) > windowWidth) continue;
      /* the reference indexes its W*H gradient array with the FLAT index (:507); the window is the ceiling of the
       * checked one, so a coordinate may reach -1, W or H: column -1 / W then reads the neighbouring row's gradient
       * (as upstream); an index outside the array is an out-of-bounds read upstream (undefined) and is DEFINED here
       * as a zero gradient, which votes nothing. */
      long long flat = llroundf(cy + ky) * (long long)W + llroundf(cx + kx);
      o_float2 g = {0.0f, 0.0f};
      if (flat >= 0 && flat < (long long)W * H) g = pixel_gradient(level, (int)W, (int)H, (int)(flat % W), (int)(flat / W));
      float mag = sqrtf((g.x * g.x) + (g.y * g.y)) * sv_expf(-((cx * cx) + (cy * cy)) / (2.0f * windowWidth * windowWidth));
      float ang = fmodf(sv_atan2f(g.y, g.x) - theta + (2.0f * pi), 2.0f * pi);
      for (float nx = 0; nx < 4.0f; nx += 1.0f) {
        for (float ny = 0; ny < 4.0f; ny += 1.0f) {
          float hx = (nx * 0.5f - 0.75f) * windowWidth, hy = (ny * 0.5f - 0.75f) * windowWidth;
          float rx = (hx * c) + (hy * s), ry = (-hx * s) + (hy * c);
          hx = fabsf(rx - cx);
          hy = fabsf(ry - cy);
          if (hx <= binWidth && hy <= binWidth) {
            hx = hx / binWidth;
            hy = hy / binWidth;
            for (int k = 0; k < 8; ++k) {
              float angle = fabsf(ang - ((float)k * rad45));
              if (angle < rad45) {
                angle /= rad45;
                float temp = (1.0f - hx) * (1.0f - hy) * (1.0f - angle) * mag;
                if (exact) ibins[(int)nx][(int)ny][k] += vote_fixed(temp * voteScale);
                else bins[(int)nx][(int)ny][k] += temp;
              }
            }
          }
        }
      }
    }
  }
  if (exact) { /* the normalisation below is scale invariant: the integer sums are used as they are */
    for (int xx = 0; xx < 4; ++xx)
      for (int yy = 0; yy < 4; ++yy)
        for (int z = 0; z < 8; ++z) bins[xx][yy][z] = (float)ibins[xx][yy][z];
  }
  float norm = 0.0f;
  if (exact) norm = tree_norm(&bins[0][0][0]);
  else
    for (int z = 0; z < 8; ++z)
      for (int yy = 0; yy < 4; ++yy)
        for (int xx = 0; xx < 4; ++xx) norm += bins[xx][yy][z] * bins[xx][yy][z];
  float sq = sqrtf(norm);
  for (int xx = 0; xx < 4; ++xx)
    for (int yy = 0; yy < 4; ++yy)
      for (int z = 0; z < 8; ++z) {
        bins[xx][yy][z] /= sq;
        if (bins[xx][yy][z] > 0.2f) bins[xx][yy][z] = 0.2f;
      }
  norm = 0.0f;
  if (exact) norm = tree_norm(&bins[0][0][0]);
  else
    for (int z = 0; z < 8; ++z)
      for (int yy = 0; yy < 4; ++yy)
        for (int xx = 0; xx < 4; ++xx) norm += bins[xx][yy][z] * bins[xx][yy][z];
  sq = sqrtf(norm);
  for (int xx = 0; xx < 4; ++xx)
    for (int yy = 0; yy < 4; ++yy)
      for (int z = 0; z < 8; ++z)
        feat->values[(yy * 4 + xx) * 8 + z] = (uint8_t)roundf(255.0f * bins[xx][yy][z] / sq);
  feat->theta = kp->theta;
  feat->sigma = kp->sigma;
  feat->loc.x = kp->loc.x * pixelWidth;
  feat->loc.y = kp->loc.y * pixelWidth;
  feat->parent = -1; /* Feature() default, never written by the kernel */
}

/* Stages: 0 raw extrema, 1 +removeNoise(0.8*thr), 2 +refine, 3 +removeNoise(thr), 4 +removeEdges,
 *         5 +checkKeyPoints, 6 +orientations (SIFT_FeatureFactory.cu:71-110, FeatureFactory.cu:461-509). */
static void run_keypoints(oracle_sift* s, int stage, unsigned int maxOrientations, float orientationThreshold,
                          float orientationContribWidth, float descriptorContribWidth) {
  float noiseThreshold = 0.01f;
  float edgeThreshold = 12.1f;
  for (int o = 0; o < NUM_OCT; ++o) {
    octave_t* oc = &s->oct[o];
    free(oc->extrema);
    oc->extrema = NULL;
    oc->n = 0;
    memset(oc->idx, 0, sizeof oc->idx);
    search_extrema(oc, o);
    if (!oc->extrema) continue;
    if (stage >= 1) remove_noise(oc, (float)(noiseThreshold * 0.8)); /* float*double narrowed at the call (:484) */
    if (stage >= 2) refine_extrema(oc);
    if (stage >= 3) remove_noise(oc, noiseThreshold);
    if (stage >= 4) remove_edges(oc, edgeThreshold);
  }
  if (stage >= 5)
    for (int o = 0; o < NUM_OCT; ++o) check_keypoints(&s->oct[o], descriptorContribWidth);
  if (stage >= 6)
    for (int o = 0; o < NUM_OCT; ++o)
      compute_orientations(&s->oct[o], orientationThreshold, maxOrientations, orientationContribWidth);
}

int oracle_sift_keypoints(oracle_sift* s, int stage, o_sskeypoint** out, int blurIndices[4][6]) {
  run_keypoints(s, stage, 2, 0.8f, 1.5f, 6.0f);
  int total = 0;
  for (int o = 0; o < NUM_OCT; ++o) total += s->oct[o].n;
  o_sskeypoint* list = (o_sskeypoint*)malloc(sizeof(o_sskeypoint) * (total ? total : 1));
  int k = 0;
  for (int o = 0; o < NUM_OCT; ++o) {
    if (s->oct[o].n) memcpy(list + k, s->oct[o].extrema, sizeof(o_sskeypoint) * s->oct[o].n);
    k += s->oct[o].n;
    if (blurIndices) {
      for (int b = 0; b < NUM_DOG; ++b) blurIndices[o][b] = s->oct[o].idx[b];
      blurIndices[o][5] = s->oct[o].n;
    }
  }
  *out = list;
  return total;
}

static int build_features(oracle_sift* s, float descriptorContribWidth, o_feature** out) {
  int total = 0;
  for (int o = 0; o < NUM_OCT; ++o) total += s->oct[o].n;
  o_feature* feats = (o_feature*)calloc(total ? total : 1, sizeof(o_feature));
  int produced = 0;
  for (int o = 0; o < NUM_OCT; ++o) {
    octave_t* oc = &s->oct[o];
    if (!oc->extrema) continue;
    for (int b = 0; b < NUM_DOG; ++b) {
      int cnt = (b + 1 == NUM_DOG) ? oc->n - oc->idx[b] : oc->idx[b + 1] - oc->idx[b];
      if (cnt <= 0) continue;
#pragma omp parallel for schedule(dynamic, 16)
      for (int k = 0; k < cnt; ++k)
        oracle_fill_descriptor(oc->dogn[b], oc->w, oc->h, oc->pixelWidth, descriptorContribWidth,
                               &oc->extrema[oc->idx[b] + k], &feats[produced + k]);
      produced += cnt;
    }
  }
  *out = feats;
  return produced;
}

int oracle_sift_features(oracle_sift* s, uint32_t maxOrientations, float orientationThreshold,
                         float orientationContribWidth, float descriptorContribWidth, o_feature** out) {
  run_keypoints(s, 6, maxOrientations, orientationThreshold, orientationContribWidth, descriptorContribWidth);
  return build_features(s, descriptorContribWidth, out);
}

int oracle_sift_generate(const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t maxOrientations,
                         float orientationThreshold, float orientationContribWidth, float descriptorContribWidth,
                         o_feature** out) {
  oracle_sift* s = oracle_sift_create(pixels, width, height);
  if (!s) return -1;
  int n = oracle_sift_features(s, maxOrientations, orientationThreshold, orientationContribWidth,
                               descriptorContribWidth, out);
  oracle_sift_destroy(s);
  return n;
}

/* kernel-level wrappers for parity tests of the individual image ops */
void oracle_upsample2x(const float* in, uint32_t w, uint32_t h, float* out) {
  float* r = upsample2x(in, w, h);
  memcpy(out, r, sizeof(float) * 4 * (size_t)w * h);
  free(r);
}
void oracle_bin2x(const float* in, uint32_t w, uint32_t h, float* out) {
  float* r = bin2x(in, w, h);
  memcpy(out, r, sizeof(float) * (size_t)(w / 2) * (h / 2));
  free(r);
}
void oracle_conv_separable(const float* in, uint32_t w, uint32_t h, int taps, const float* weights, float* out) {
  float* r = conv_separable(in, w, h, taps, weights);
  memcpy(out, r, sizeof(float) * (size_t)w * h);
  free(r);
}

/* generateBW (src/Image.cu:1277-1296) with bwaToBW / rgbToBW / rgbaToBW / rgbaToRGB (:1253-1275): the operands are
 * unsigned chars promoted to int, the result is narrowed to unsigned char */
void oracle_convert_to_bw(const uint8_t* color, uint32_t colorDepth, uint8_t* bw, size_t numPixels) {
  for (size_t i = 0; i < numPixels; ++i) {
    const uint8_t* p = color + i * colorDepth;
    if (colorDepth == 2) {
      bw[i] = (uint8_t)((1 - p[1]) * p[0] + p[1] * p[0]);
    } else if (colorDepth == 3) {
      bw[i] = (uint8_t)((p[0] / 4) + (p[1] / 2) + (p[2] / 4));
    } else {
      uint8_t r = (uint8_t)((1 - p[3]) * p[0] + p[3] * p[0]);
      uint8_t g = (uint8_t)((1 - p[3]) * p[1] + p[3] * p[1]);
      uint8_t b = (uint8_t)((1 - p[3]) * p[2] + p[3] * p[2]);
      bw[i] = (uint8_t)((r / 4) + (g / 2) + (b / 4));
    }
  }
}
