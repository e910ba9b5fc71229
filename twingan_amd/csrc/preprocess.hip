// Training-image preprocessing on the GPU: decoded uint8 images of arbitrary size -> the [n, hw, hw, 3] batch in [0, 1]
// the networks train on.  One launch per batch, one thread per output pixel.
//
// Reference call site replaced: preprocessing/danbooru_preprocessing.py:115-230 (preprocess_image, the factory entry
// for the TwinGAN trainer: model/model_inheritor.py:403-457) with the trainer's defaults -- dtype conversion to [0, 1]
// (tf.image.convert_image_dtype), resize_mode PAD / CROP / RESHAPE to a square (preprocessing_util.py:97-146:
// pad_to_bounding_box / crop_to_bounding_box about the centre, then tf.image.resize_images BILINEAR,
// align_corners=False = the TF-1.x kernel without half-pixel centres), random_flip_left_right
// (preprocessing_util.py:171-205), distort_color in fast mode (danbooru_preprocessing.py:62-113: random_brightness
// max_delta 32/255 and random_saturation [0.5, 1.5) in one of two orders), tf.clip_by_value(0, 1).  The random draws are
// inputs (aug[n][4] = flip?, saturation first?, brightness delta, saturation factor): the host draws them.
//
// --do_random_cropping (model_inheritor.py:225,449-454; the reference's training recipe sets it, docs/training.md:22-23;
// danbooru_preprocessing.py:187-201, preprocessing_util.random_crop_image :312-331): the first bilinear resize goes to
// mid = int(hw / 0.8), tf.random_crop cuts a [ch, cw] rectangle out of that at (cy, cx) (crop[n][4] = cy, cx, ch, cw, drawn
// by the host), a second bilinear resize brings the rectangle to [hw, hw].  The kernel never materialises the mid x mid
// image: an output pixel's four taps in the rectangle are each evaluated from their four source taps (16 fetches).
// --color_space (model_inheritor.py:240,414; danbooru_preprocessing.py:208-225): 'gray' skips the colour distortion,
// 'yiq' / 'bgr' transform the finished image (preprocessing_util.rgb_to_yiq :154-160, tf.reverse on the channel axis).
#include "tg_common.h"

namespace {

struct PreGeom {
  int n, hw, mid, color_space;      // mid: side of the intermediate image (with a crop table); TG_CS_*
};

enum { TG_CS_RGB = 0, TG_CS_YIQ = 1, TG_CS_BGR = 2, TG_CS_GRAY = 3 };

__device__ __forceinline__ float3 fetch(const uint8_t* img, int h, int w, int y0, int x0, int vy, int vx) {
  // virtual source pixel (vy, vx) -> image pixel (vy + y0, vx + x0); outside the image: the zero padding
  const int y = vy + y0, x = vx + x0;
  if (y < 0 || y >= h || x < 0 || x >= w) return make_float3(0.f, 0.f, 0.f);
  const uint8_t* p = img + ((int64_t)y * w + x) * 3;
  const float k = 1.0f / 255.0f;      // tf.image.convert_image_dtype(uint8 -> float32): cast * (1 / max)
  return make_float3((float)p[0] * k, (float)p[1] * k, (float)p[2] * k);
}

__device__ __forceinline__ float3 lerp3(float3 a, float3 b, float t) {
  return make_float3(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t);
}

// tf.image.adjust_saturation: RGB -> HSV, s = clip(s * factor, 0, 1), HSV -> RGB.  Hue and value do not change, and
// v - channel = s * v * (1 - d_channel(h)) is linear in s, so the round trip is  v - (v - channel) * (s' / s).
__device__ __forceinline__ float3 saturate(float3 c, float factor) {
  const float v = fmaxf(c.x, fmaxf(c.y, c.z));
  const float range = v - fminf(c.x, fminf(c.y, c.z));
  if (!(v > 0.f) || !(range > 0.f)) return c;          // s = 0: grey stays grey
  const float s = range / v;
  const float ratio = fminf(s * factor, 1.f) / s;
  return make_float3(v - (v - c.x) * ratio, v - (v - c.y) * ratio, v - (v - c.z) * ratio);
}

struct Src {      // one decoded image and the rectangle of it (in image coordinates) that the first resize reads
  const uint8_t* img;
  int h, w, y0, x0, sh, sw;
};

// pixel (oy, ox) of ResizeBilinear(source rectangle -> [size, size]), align_corners = False: in = out * (in_size / out_size)
__device__ __forceinline__ float3 resized(const Src& s, int oy, int ox, float sy, float sx) {
  const float fy = (float)oy * sy, fx = (float)ox * sx;
  const int top = (int)floorf(fy), left = (int)floorf(fx);
  const int bot = min(top + 1, s.sh - 1), right = min(left + 1, s.sw - 1);
  const float ly = fy - (float)top, lx = fx - (float)left;
  const float3 t = lerp3(fetch(s.img, s.h, s.w, s.y0, s.x0, top, left), fetch(s.img, s.h, s.w, s.y0, s.x0, top, right), lx);
  const float3 b = lerp3(fetch(s.img, s.h, s.w, s.y0, s.x0, bot, left), fetch(s.img, s.h, s.w, s.y0, s.x0, bot, right), lx);
  return lerp3(t, b, ly);
}

template <typename T, bool CROP>
__global__ void preprocess_kernel(const uint8_t* __restrict__ packed, const int64_t* __restrict__ offsets,
                                  const int* __restrict__ rect, const int* __restrict__ crop,
                                  const float* __restrict__ aug, T* __restrict__ out, PreGeom g) {
  const int n = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.hw * g.hw) return;
  int oy = idx / g.hw, ox = idx - oy * g.hw;
  const int* r = rect + n * 6;                         // image h, w; source rectangle y0, x0, sh, sw
  Src s;
  s.img = packed + offsets[n];
  s.h = r[0], s.w = r[1], s.y0 = r[2], s.x0 = r[3], s.sh = r[4], s.sw = r[5];
  const float* a = aug + n * 4;
  if (a[0] != 0.f) ox = g.hw - 1 - ox;                 // tf.reverse(image, [1]) of the RESIZED image
  float3 c;
  if (CROP) {
    const int* cr = crop + n * 4;                      // the rectangle tf.random_crop cut out of the mid x mid image
    const int cy = cr[0], cx = cr[1], ch = cr[2], cw = cr[3];
    const float sy = (float)s.sh / (float)g.mid, sx = (float)s.sw / (float)g.mid;
    const float fy = (float)oy * ((float)ch / (float)g.hw), fx = (float)ox * ((float)cw / (float)g.hw);
    const int top = (int)floorf(fy), left = (int)floorf(fx);
    const int bot = min(top + 1, ch - 1), right = min(left + 1, cw - 1);
    const float ly = fy - (float)top, lx = fx - (float)left;
    const float3 t = lerp3(resized(s, cy + top, cx + left, sy, sx), resized(s, cy + top, cx + right, sy, sx), lx);
    const float3 b = lerp3(resized(s, cy + bot, cx + left, sy, sx), resized(s, cy + bot, cx + right, sy, sx), lx);
    c = lerp3(t, b, ly);
  } else {
    c = resized(s, oy, ox, (float)s.sh / (float)g.hw, (float)s.sw / (float)g.hw);
  }
  if (g.color_space != TG_CS_GRAY) {                   // danbooru_preprocessing.py:208-212: no distort_color for 'gray'
    const float delta = a[2], factor = a[3];
    if (a[1] == 0.f) {                                 // ordering 0: brightness, then saturation
      c = make_float3(c.x + delta, c.y + delta, c.z + delta);
      c = saturate(c, factor);
    } else {                                           // orderings 1-3 (fast mode): saturation, then brightness
      c = saturate(c, factor);
      c = make_float3(c.x + delta, c.y + delta, c.z + delta);
    }
    c = make_float3(fminf(fmaxf(c.x, 0.f), 1.f), fminf(fmaxf(c.y, 0.f), 1.f), fminf(fmaxf(c.z, 0.f), 1.f));
  }
  if (g.color_space == TG_CS_YIQ) {                    // preprocessing_util.rgb_to_yiq: tensordot with the fp32 matrix
    c = make_float3(0.299f * c.x + 0.587f * c.y + 0.114f * c.z, 0.596f * c.x - 0.274f * c.y - 0.322f * c.z,
                    0.211f * c.x - 0.523f * c.y + 0.312f * c.z);
  } else if (g.color_space == TG_CS_BGR) {
    c = make_float3(c.z, c.y, c.x);
  }
  T* o = out + (((int64_t)n * g.hw + oy) * g.hw + (a[0] != 0.f ? g.hw - 1 - ox : ox)) * 3;
  st(o + 0, c.x);
  st(o + 1, c.y);
  st(o + 2, c.z);
}

}  // namespace

extern "C" int tg_preprocess_images_crop(const void* packed, const int64_t* offsets, const int* rect, const int* crop,
                                         const float* aug, void* out, int n, int hw, int mid, int color_space, int dtype,
                                         void* stream) {
  TG_CHECK(packed && offsets && rect && aug && out && n > 0 && hw > 0, TG_EINVAL, "tg_preprocess_images: bad arguments");
  TG_CHECK(color_space >= TG_CS_RGB && color_space <= TG_CS_GRAY, TG_EINVAL, "tg_preprocess_images: color_space 0..3");
  TG_CHECK(!crop || mid >= hw, TG_EINVAL, "tg_preprocess_images: a crop table needs the intermediate size mid >= hw");
  PreGeom g;
  g.n = n;
  g.hw = hw;
  g.mid = mid;
  g.color_space = color_space;
  const dim3 grid((hw * hw + 255) / 256, n);
  TG_DISPATCH_DTYPE(dtype, "tg_preprocess_images", {
    if (crop)
      hipLaunchKernelGGL((preprocess_kernel<T, true>), grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)packed,
                         offsets, rect, crop, aug, (T*)out, g);
    else
      hipLaunchKernelGGL((preprocess_kernel<T, false>), grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)packed,
                         offsets, rect, crop, aug, (T*)out, g);
  });
  TG_LAUNCH_CHECK("tg_preprocess_images");
  return TG_OK;
}

extern "C" int tg_preprocess_images(const void* packed, const int64_t* offsets, const int* rect, const float* aug, void* out,
                                    int n, int hw, int dtype, void* stream) {
  return tg_preprocess_images_crop(packed, offsets, rect, nullptr, aug, out, n, hw, 0, TG_CS_RGB, dtype, stream);
}
