// Linear-BVH construction on the current HIP device, same output as buildBvh (bvh.h).
#pragma once
#include "bvh.h"
#include <string>

namespace pt {

// v = T x 9 floats on the host.  Returns false with a reason when the tree is not usable (fewer than two hittable
// triangles, everything fits one leaf, radix tree deeper than the traversal stack, HIP error): the caller then
// builds on the host.
bool buildBvhGpu(const float* v, uint32_t T, uint32_t maxLeaf, Bvh& out, std::string& why);

}  // namespace pt
