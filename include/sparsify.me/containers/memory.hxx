// memory.hxx -- where a container's storage lives.
// Mirrors the reference's include/sparsify.me/containers/memory.hxx:13 (same enumerators).
#pragma once
namespace sparsifyme {
enum memory_space_t { device, host };
}  // namespace sparsifyme
