"""Build the reference's OWN device kernels into oracle/_ref/libmqdet_ref.so (test infrastructure only).

The reference's native ops on the hot path exist only as CUDA sources:
    maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu   dmcn_im2col_bilinear (:474-504),
                                                              modulated_deformable_im2col_gpu_kernel (:577-640)
    maskrcnn_benchmark/csrc/cuda/ml_nms.cu                    devIoU + ml_nms_kernel (:13-75)
    maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu             bilinear_interpolate + RoIAlignForward (:16-123)
    groundingdino_new/.../MsDeformAttn/ms_deform_im2col_cuda.cuh   ms_deform_attn_im2col_bilinear (:33-84),
                                                              ms_deformable_im2col_gpu_kernel (:237-299)
Their host wrappers need ATen / THC CUDA headers and cannot be built here, but the DEVICE code is plain CUDA C.  This
script reads those line ranges from the sources WHERE THEY LIE under /root/reference (nothing is copied into the
repository: the extracted text only ever exists in a temporary file that is deleted after compilation), puts a
12-line prelude in front (the two loop macros the files define themselves, min/max, at::ceil_div), appends extern "C"
launchers written here that follow the reference's host wrappers (grid / block sizes cited below), and compiles the
result with hipcc for gfx950 into oracle/_ref/libmqdet_ref.so (git-ignored; travels to the GPU box with the snapshot).

Used by tests/ (GPU) to PIN oracle.head.dcn_v2 / oracle.postprocess.ml_nms / oracle.roi.roi_align /
oracle.gdino.ms_deform_attn_core and the HIP kernels against what the reference's device code computes.
Never imported by mq_det_amd.
"""
import os
import subprocess
import sys
import tempfile

REF = os.environ.get("MQDET_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libmqdet_ref.so")

# (file, first line, last line, text the first line must contain) -- 1-based, inclusive
RANGES = [
    ("maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu", 474, 504, "template <typename scalar_t>"),
    ("maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu", 577, 640, "template <typename scalar_t>"),
    ("maskrcnn_benchmark/csrc/cuda/ml_nms.cu", 13, 75, "int const threadsPerBlock"),
    ("maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu", 16, 123, "template <typename T>"),
    ("groundingdino_new/models/GroundingDINO/csrc_groundingdino/MsDeformAttn/ms_deform_im2col_cuda.cuh", 33, 84,
     "template <typename scalar_t>"),
    ("groundingdino_new/models/GroundingDINO/csrc_groundingdino/MsDeformAttn/ms_deform_im2col_cuda.cuh", 237, 299,
     "template <typename scalar_t>"),
]

PRELUDE = r'''
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <algorithm>
// the two grid-stride loop macros are defined by the reference files themselves (deform_conv_kernel_cuda.cu:70-72,
// ROIAlign_cuda.cu:11-13, ms_deform_im2col_cuda.cuh:20-23); at::ceil_div is ATen's integer ceil-div (ml_nms.cu:71)
#define CUDA_KERNEL_LOOP(i, n) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)
#define CUDA_1D_KERNEL_LOOP(i, n) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.x)
namespace at { template <typename T> __host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; } }
'''

LAUNCHERS = r'''
// ---- extern "C" launchers (written for this repository; they mirror the reference's host wrappers) -------------
extern "C" int ref_abi_version() { return 1; }

// modulated_deformable_im2col_cuda (deform_conv_kernel_cuda.cu:757-789): num_kernels = C * B * Hcol * Wcol,
// GET_BLOCKS(n) = min(65535, ceil(n / 1024)) blocks of 1024 threads (:74-80).  kernel 3x3, dilation 1, one deformable group.
extern "C" int ref_dcn_im2col(const float* im, const float* offset, const float* mask, float* col, int B, int C, int H, int W,
                              int Hcol, int Wcol, int pad, int stride, void* stream) {
  const int n = C * B * Hcol * Wcol;
  const int blocks = std::min(65535, (n + 1023) / 1024);
  hipLaunchKernelGGL((modulated_deformable_im2col_gpu_kernel<float>), dim3(blocks), dim3(1024), 0, (hipStream_t)stream, n, im,
                     offset, mask, H, W, 3, 3, pad, pad, stride, stride, 1, 1, C, B, C, 1, Hcol, Wcol, col);
  return (int)hipGetLastError();
}

// ml_nms_cuda (ml_nms.cu:78-149): boxes [n, 6] = (x1, y1, x2, y2, score, label) ALREADY sorted by score (the wrapper
// sorts first, :82-84); 64-thread blocks on a col_blocks x col_blocks grid (:103-110); then the serial host sweep
// (:122-141), restated here line by line.  keep[i] = 1 for kept boxes of the sorted order.  Returns #kept.
extern "C" int ref_ml_nms(const float* boxes_sorted_dev, int n, float thresh, unsigned char* keep_host) {
  if (n <= 0) return 0;
  const int col_blocks = (n + threadsPerBlock - 1) / threadsPerBlock;
  unsigned long long* mask_dev = nullptr;
  if (hipMalloc(&mask_dev, sizeof(unsigned long long) * (size_t)n * col_blocks) != hipSuccess) return -1;
  hipLaunchKernelGGL(ml_nms_kernel, dim3(col_blocks, col_blocks), dim3(threadsPerBlock), 0, 0, n, thresh, boxes_sorted_dev, mask_dev);
  unsigned long long* mask_host = (unsigned long long*)malloc(sizeof(unsigned long long) * (size_t)n * col_blocks);
  hipMemcpy(mask_host, mask_dev, sizeof(unsigned long long) * (size_t)n * col_blocks, hipMemcpyDeviceToHost);
  unsigned long long* remv = (unsigned long long*)calloc(col_blocks, sizeof(unsigned long long));
  int num_to_keep = 0;
  for (int i = 0; i < n; i++) {
    keep_host[i] = 0;
    int nblock = i / threadsPerBlock, inblock = i % threadsPerBlock;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep_host[i] = 1;
      ++num_to_keep;
      unsigned long long* p = mask_host + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
    }
  }
  free(remv); free(mask_host); hipFree(mask_dev);
  return num_to_keep;
}

// ROIAlign_forward_cuda (ROIAlign_cuda.cu:262-306): output_size = R * C * PH * PW, grid = min(ceil(n / 512), 4096), 512 threads.
extern "C" int ref_roi_align(const float* feat, const float* rois, float* out, int R, int C, int H, int W, int PH, int PW,
                             float spatial_scale, int sampling_ratio, void* stream) {
  const int n = R * C * PH * PW;
  if (n == 0) return 0;
  const int blocks = std::min((n + 511) / 512, 4096);
  hipLaunchKernelGGL((RoIAlignForward<float>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, n, feat, spatial_scale, C, H, W,
                     PH, PW, sampling_ratio, rois, out);
  return (int)hipGetLastError();
}

// ms_deformable_im2col_cuda (ms_deform_im2col_cuda.cuh:924-953): n = B * Q * heads * channels, 1024-thread blocks.
// value [B, S, heads, ch], spatial_shapes [L, 2] int64, level_start [L] int64, loc [B, Q, heads, L, P, 2], attn [B, Q, heads, L, P]
extern "C" int ref_ms_deform_im2col(const float* value, const int64_t* shapes, const int64_t* level_start, const float* loc,
                                    const float* attn, float* col, int B, int S, int heads, int ch, int L, int Q, int P, void* stream) {
  const int n = B * Q * heads * ch;
  if (n == 0) return 0;
  hipLaunchKernelGGL((ms_deformable_im2col_gpu_kernel<float>), dim3((n + 1023) / 1024), dim3(1024), 0, (hipStream_t)stream, n,
                     value, shapes, level_start, loc, attn, B, S, heads, ch, L, Q, P, col);
  return (int)hipGetLastError();
}
'''


def extract():
    parts = []
    for rel, a, b, must in RANGES:
        path = os.path.join(REF, rel)
        lines = open(path).read().split("\n")
        if must not in lines[a - 1]:
            raise RuntimeError(f"{rel}:{a} does not start with '{must}' (reference layout changed?): {lines[a - 1]!r}")
        parts.append(f"// ---- {rel}:{a}-{b}\n" + "\n".join(lines[a - 1:b]) + "\n")
    return "\n".join(parts)


def build(force=False):
    """-> path of the .so, or None when neither the reference sources nor a prebuilt library are present."""
    if not os.path.isdir(REF):
        return OUT if os.path.exists(OUT) else None          # GPU box: use the prebuilt file that travelled
    os.makedirs(OUT_DIR, exist_ok=True)
    stamp = os.path.getmtime(os.path.abspath(__file__))
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= stamp:
        return OUT
    src = PRELUDE + extract() + LAUNCHERS
    with tempfile.TemporaryDirectory(prefix="mqdet_ref_") as tmp:     # extracted reference text never lands in the repo
        f = os.path.join(tmp, "ref_kernels.hip")
        open(f, "w").write(src)
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-value", f, "-o", OUT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on the extracted reference kernels:\n" + r.stderr[-4000:])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
