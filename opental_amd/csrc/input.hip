// opental_amd/csrc/input.hip -- clip preparation on the device.
//
// Replaces the host-side tail of THUMOS_Dataset.__getitem__ (AFSD/common/thumos_dataset.py:246-262) and the transforms
// it calls (AFSD/common/videotransforms.py:44-124): temporal slice + zero padding to clip_length, random / centre crop,
// horizontal flip, uint8 -> float, (x / 255) * 2 - 1, and the (T,H,W,3) -> (3,T,H,W) transpose of load_video_data
// (thumos_dataset.py:136-137).  The reference does all of it in numpy per sample and ships a 28 MB fp32 clip to the
// GPU; here the uint8 frames (4x smaller) are uploaded and one launch writes the normalised batch.  The random crop
// offsets and flip decisions stay on the host (same `random` calls, see common/input_pipeline.py) and arrive as
// parameters, so results are reproducible against the reference bit for bit.
// HBM-bound: 3 bytes read + 12 bytes written per pixel.
#include "common.h"

namespace {

struct ClipParams {      // one per clip, device array
    long long frame0;    // element offset of this clip's first frame in the uint8 frame buffer
    int valid_t;         // frames available (the rest of the clip is zero BEFORE normalisation, i.e. -1.0 after it)
    int crop_i, crop_j;  // top-left corner of the crop in the source frame
    int flip;            // bit 0: mirror along W after cropping; bit 1: pad frames are 127.5 before normalisation (exactly
                         // 0.0 after it: the ActivityNet loader, AFSD/common/anet_dataset.py:226-229) instead of 0 (-1.0)
};

__global__ __launch_bounds__(256) void prepare_clips_kernel(const unsigned char* __restrict__ frames,
                                                            const ClipParams* __restrict__ params, float* __restrict__ out,
                                                            int T, int Hs, int Ws, int Ho, int Wo) {
    const int b = blockIdx.y;
    const ClipParams p = params[b];
    const int plane = Ho * Wo;
    const long long vol = (long long)T * plane;
    float* ob = out + (long long)b * 3 * vol;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < vol; idx += (long long)gridDim.x * 256) {
        const int t = (int)(idx / plane);
        const int r = (int)(idx - (long long)t * plane);
        const int y = r / Wo, x = r - y * Wo;
        float v0, v1, v2;
        if (t < p.valid_t) {
            const int xs = p.crop_j + ((p.flip & 1) ? Wo - 1 - x : x);
            const unsigned char* px = frames + p.frame0 + (((long long)t * Hs + (p.crop_i + y)) * Ws + xs) * 3;
            v0 = ((float)px[0] / 255.0f) * 2.0f - 1.0f;     // same operation order as the reference (no fma: -ffp-contract=off)
            v1 = ((float)px[1] / 255.0f) * 2.0f - 1.0f;
            v2 = ((float)px[2] / 255.0f) * 2.0f - 1.0f;
        } else {
            v0 = v1 = v2 = (((p.flip & 2) ? 127.5f : 0.0f) / 255.0f) * 2.0f - 1.0f;
        }
        ob[idx] = v0;
        ob[vol + idx] = v1;
        ob[2 * vol + idx] = v2;
    }
}

// The same preparation with an optional FRAME MAP per clip: output frame t is source frame map[t] of the (padded) clip.
// The self-supervised branch's spliced clip (thumos_dataset.py:187-228 `augment_`: two time segments of the normalised clip
// swap places) is exactly that, so one launch writes the plain batch and -- from the same uint8 upload and the same crop /
// flip decisions -- the ssl batch; the reference clones and slices a 28 MB fp32 clip per sample on the host.
__global__ __launch_bounds__(256) void prepare_clips_map_kernel(const unsigned char* __restrict__ frames,
                                                                const ClipParams* __restrict__ params,
                                                                const int* __restrict__ fmap, float* __restrict__ out,
                                                                float* __restrict__ out_ssl, int T, int Hs, int Ws, int Ho, int Wo) {
    const int b = blockIdx.y;
    const ClipParams p = params[b];
    const int plane = Ho * Wo;
    const long long vol = (long long)T * plane;
    for (int pass = 0; pass < 2; ++pass) {
        float* dst = pass == 0 ? out : out_ssl;
        if (!dst) continue;
        float* ob = dst + (long long)b * 3 * vol;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < vol; idx += (long long)gridDim.x * 256) {
            const int t = (int)(idx / plane);
            const int r = (int)(idx - (long long)t * plane);
            const int y = r / Wo, x = r - y * Wo;
            const int ts = (pass == 1 && fmap) ? fmap[b * T + t] : t;
            float v0, v1, v2;
            if (ts >= 0 && ts < p.valid_t) {
                const int xs = p.crop_j + ((p.flip & 1) ? Wo - 1 - x : x);
                const unsigned char* px = frames + p.frame0 + (((long long)ts * Hs + (p.crop_i + y)) * Ws + xs) * 3;
                v0 = ((float)px[0] / 255.0f) * 2.0f - 1.0f;
                v1 = ((float)px[1] / 255.0f) * 2.0f - 1.0f;
                v2 = ((float)px[2] / 255.0f) * 2.0f - 1.0f;
            } else {
                v0 = v1 = v2 = (((p.flip & 2) ? 127.5f : 0.0f) / 255.0f) * 2.0f - 1.0f;
            }
            ob[idx] = v0;
            ob[vol + idx] = v1;
            ob[2 * vol + idx] = v2;
        }
    }
}

struct WindowParams {    // one per inference window, device array (16 bytes)
    unsigned long long src;  // device address of frame `offset` of channel 0 of the window's video (planar uint8 (C,Tv,H,W))
    int chan_stride4;        // Tv * H * W / 4: distance between the channels of that video, in 4-byte words
    int valid_t;             // frames available from `offset`; the rest of the window is 0.0 (zero padding AFTER normalisation)
};

// 4 pixels per thread: one dword load, one 16-byte store
__global__ __launch_bounds__(256) void prepare_windows_kernel(const WindowParams* __restrict__ params, float4* __restrict__ out,
                                                              int C, int T, int plane4) {
    const int b = blockIdx.y;
    const WindowParams p = params[b];
    const long long vol4 = (long long)T * plane4;
    const unsigned* src = reinterpret_cast<const unsigned*>(p.src);
    const long long valid4 = (long long)p.valid_t * plane4;
    for (int c = 0; c < C; ++c) {
        float4* ob = out + ((long long)b * C + c) * vol4;
        const unsigned* sc = src + (long long)c * p.chan_stride4;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < vol4; idx += (long long)gridDim.x * 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < valid4) {
                const unsigned w = sc[idx];
                // the reference normalises the windows ON THE GPU (test.py:67-76 on a cuda tensor), where torch divides by a
                // scalar as a multiplication by its fp32 reciprocal -- unlike the training transforms, which run in numpy
                const float inv = 1.0f / 255.0f;
                v.x = ((float)(w & 0xffu) * inv) * 2.0f - 1.0f;
                v.y = ((float)((w >> 8) & 0xffu) * inv) * 2.0f - 1.0f;
                v.z = ((float)((w >> 16) & 0xffu) * inv) * 2.0f - 1.0f;
                v.w = ((float)(w >> 24) * inv) * 2.0f - 1.0f;
            }
            ob[idx] = v;
        }
    }
}

}  // namespace

extern "C" int otal_prepare_windows(const void* params, float* out, int B, int C, int T, int H, int W, void* stream) {
    if (!params || !out) return OTAL_E_NULL;
    if (B <= 0 || C <= 0 || T <= 0 || H <= 0 || W <= 0) return OTAL_E_SHAPE;
    if (B > 65535 || ((long long)H * W) % 4) return OTAL_E_UNSUPPORTED;
    const int plane4 = H * W / 4;
    const long long vol4 = (long long)T * plane4;
    const int bx = (int)((vol4 + 255) / 256 < 2048 ? (vol4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(prepare_windows_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const WindowParams*>(params), reinterpret_cast<float4*>(out), C, T, plane4);
    return otal_launch_status();
}

extern "C" int otal_prepare_clips(const unsigned char* frames, const void* params, float* out, int B, int T, int Hs,
                                  int Ws, int Ho, int Wo, void* stream) {
    if (!frames || !params || !out) return OTAL_E_NULL;
    if (B <= 0 || T <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || Ho > Hs || Wo > Ws) return OTAL_E_SHAPE;
    if (B > 65535) return OTAL_E_UNSUPPORTED;
    const long long vol = (long long)T * Ho * Wo;
    const int bx = (int)((vol + 255) / 256 < 4096 ? (vol + 255) / 256 : 4096);
    hipLaunchKernelGGL(prepare_clips_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, frames,
                       static_cast<const ClipParams*>(params), out, T, Hs, Ws, Ho, Wo);
    return otal_launch_status();
}

extern "C" int otal_prepare_clips_map(const unsigned char* frames, const void* params, const int* frame_map, float* out,
                                      float* out_ssl, int B, int T, int Hs, int Ws, int Ho, int Wo, void* stream) {
    if (!frames || !params || (!out && !out_ssl)) return OTAL_E_NULL;
    if (out_ssl && !frame_map) return OTAL_E_NULL;
    if (B <= 0 || T <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || Ho > Hs || Wo > Ws) return OTAL_E_SHAPE;
    if (B > 65535) return OTAL_E_UNSUPPORTED;
    const long long vol = (long long)T * Ho * Wo;
    const int bx = (int)((vol + 255) / 256 < 4096 ? (vol + 255) / 256 : 4096);
    hipLaunchKernelGGL(prepare_clips_map_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, frames,
                       static_cast<const ClipParams*>(params), frame_map, out, out_ssl, T, Hs, Ws, Ho, Wo);
    return otal_launch_status();
}
