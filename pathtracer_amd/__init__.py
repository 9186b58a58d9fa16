"""pathtracer_amd: MI355X-native integrator + ray-scene intersection of Ingrater/PathTracer.

The compute path is pathtracer_amd/libptgpu.so (HIP kernels for gfx950 behind the C ABI of
include/ptgpu.h); this package is the thin host-side mirror of the reference's interface."""
from .api import (PathTracer, Camera, PtError, PIXEL_DTYPE, MATERIAL_DTYPE, FILL_MATERIAL, FILL_MATERIAL_CITYMAP,
                  default_sun_dir, default_sky_radiance, pixels_from_accumulator, GpuGroup, band_rows)
from ._lib import (PtConfig, PtStats, PT_FLAG_COUNT_VISITS, PT_FLAG_NO_LDS_SCENE, PT_FLAG_TIME_KERNELS,
                   PT_FLAG_NO_COVERAGE_MASK, PT_FLAG_GPU_BVH, PT_FLAG_HOST_BVH, PT_FLAG_NO_GRAPH, PT_FLAG_NO_FUSE, PT_FLAG_QUEUED_VIEW_RAYS, PT_FLAG_NO_TREELETS, PT_FLAG_STATIC_RUNS, PT_FLAG_NO_PIXEL_SUMS, PT_FLAG_NO_HIT_ORDER, PT_FLAG_HIT_ORDER, PT_K_NAMES)

__all__ = ['PathTracer', 'Camera', 'PtError', 'PIXEL_DTYPE', 'MATERIAL_DTYPE', 'FILL_MATERIAL', 'FILL_MATERIAL_CITYMAP',
           'default_sun_dir', 'default_sky_radiance', 'pixels_from_accumulator', 'GpuGroup', 'band_rows', 'PtConfig', 'PtStats',
           'PT_FLAG_COUNT_VISITS', 'PT_FLAG_NO_LDS_SCENE', 'PT_FLAG_TIME_KERNELS', 'PT_FLAG_NO_COVERAGE_MASK', 'PT_FLAG_GPU_BVH', 'PT_FLAG_HOST_BVH', 'PT_FLAG_NO_GRAPH', 'PT_FLAG_NO_FUSE', 'PT_FLAG_QUEUED_VIEW_RAYS', 'PT_FLAG_NO_TREELETS', 'PT_FLAG_STATIC_RUNS', 'PT_FLAG_NO_PIXEL_SUMS', 'PT_FLAG_NO_HIT_ORDER', 'PT_FLAG_HIT_ORDER', 'PT_K_NAMES']
