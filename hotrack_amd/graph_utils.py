"""Introspection of captured HIP graphs (bench / profiling helpers; nothing on the data path)."""
from __future__ import annotations

import ctypes

_HIP_GRAPH_NODE_TYPE_KERNEL = 0  # hipGraphNodeTypeKernel


def _loaded_hip_runtime():
    """The HIP runtime this process ALREADY uses (torch's), by the path it is mapped from -- dlopen by a bare soname could
    bring in a second copy (the system's next to the one torch bundles), which would know nothing of torch's graphs."""
    try:
        for line in open("/proc/self/maps"):
            path = line.split()[-1]
            if "libamdhip64.so" in path:
                return ctypes.CDLL(path)
    except OSError:
        pass
    return None


def kernel_nodes(graph) -> int | None:
    """Number of kernel nodes of a torch.cuda.CUDAGraph created with keep_graph=True (= launches per replay), or None when
    the raw graph is not available."""
    try:
        raw = graph.raw_cuda_graph()
    except Exception:
        return None
    try:
        hip = _loaded_hip_runtime()
        if hip is None:
            return None
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0 or n.value == 0:
            return None
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
            return None
        count = 0
        for i in range(n.value):
            t = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) == 0 and t.value == _HIP_GRAPH_NODE_TYPE_KERNEL:
                count += 1
        return count
    except (OSError, AttributeError):
        return None
