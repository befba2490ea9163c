"""hipEvent_t handles (created by libsgx.so on its own HIP runtime) for timing launches on the
stream they run on; the library records them itself around the aggregation stage, see
sgx_layer_desc.ev_agg_begin / ev_agg_end in include/sgx.h."""
import ctypes

from ._lib import check, lib


class Event:
    def __init__(self):
        h = ctypes.c_void_p()
        check(lib.sgx_event_create(ctypes.byref(h)), "sgx_event_create")
        self.handle = h

    def record(self, stream_handle):
        check(lib.sgx_event_record(self.handle, ctypes.c_void_p(stream_handle)), "sgx_event_record")

    def elapsed_ms(self, end):
        ms = ctypes.c_float()
        check(lib.sgx_event_elapsed_ms(self.handle, end.handle, ctypes.byref(ms)), "sgx_event_elapsed_ms")
        return ms.value

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            lib.sgx_event_destroy(h)
