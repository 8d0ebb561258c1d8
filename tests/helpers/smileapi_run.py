#!/usr/bin/env python3
"""usage: smileapi_run.py <libSMILEapi.so> <conf> <pcm.npy> <out.npy> [--block SAMPLES] [--timing]
Feeds 16-bit PCM into a running openSMILE instance through the reference's C API (progsrc/include/smileapi/SMILEapi.h:
smile_new / smile_initialize / smile_run on a second thread, smile_extaudiosource_write_data in 50 ms pieces, then
smile_extaudiosource_set_external_eoi) and collects what the cExternalSink instance `extsink` hands to its callback.
Whether the plugin takes part is decided by the process's working directory (./plugins, componentManager.cpp:347-364)."""
import ctypes as C
import sys
import threading
import time

import numpy as np


class Opt(C.Structure):
    _fields_ = [("name", C.c_char_p), ("value", C.c_char_p)]


def main():
    lib, conf, pcm_path, out_path = sys.argv[1:5]
    extra = sys.argv[5:]
    step = int(extra[extra.index("--block") + 1]) if "--block" in extra else 800      # samples per write (default 50 ms)
    timing = "--timing" in extra
    L = C.CDLL(lib)
    L.smile_new.restype = C.c_void_p
    L.smile_initialize.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.smile_run.argtypes = [C.c_void_p]
    L.smile_free.argtypes = [C.c_void_p]
    L.smile_error_msg.argtypes = [C.c_void_p]
    L.smile_error_msg.restype = C.c_char_p
    L.smile_extaudiosource_write_data.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    L.smile_extaudiosource_set_external_eoi.argtypes = [C.c_void_p, C.c_char_p]
    CB = C.CFUNCTYPE(C.c_bool, C.POINTER(C.c_float), C.c_long, C.c_void_p)
    L.smile_extsink_set_data_callback.argtypes = [C.c_void_p, C.c_char_p, CB, C.c_void_p]
    pcm = np.load(pcm_path).astype(np.int16)
    rows = []

    def on_vector(data, n, _param):
        rows.append(C.string_at(data, 4 * n))
        return True
    cb = CB(on_vector)
    t_init = time.time()
    obj = L.smile_new()
    rc = L.smile_initialize(obj, conf.encode(), 0, None, 1, 0, 0, None)
    if rc != 0:
        raise SystemExit(f"smile_initialize: {rc}: {L.smile_error_msg(obj)}")
    rc = L.smile_extsink_set_data_callback(obj, b"extsink", cb, None)
    if rc != 0:
        raise SystemExit(f"smile_extsink_set_data_callback: {rc}: {L.smile_error_msg(obj)}")
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("rc", L.smile_run(obj)))
    th.start()
    pos = 0
    t0 = time.time()
    while pos < len(pcm):
        chunk = np.ascontiguousarray(pcm[pos:pos + step])
        rc = L.smile_extaudiosource_write_data(obj, b"extsource", chunk.ctypes.data, chunk.nbytes)
        if rc == 0:
            pos += len(chunk)
        else:                                             # SMILE_NOT_WRITTEN: the component's buffer is full, or the run has not started yet
            time.sleep(0.001)
            if time.time() - t0 > 240:
                raise SystemExit("smile_extaudiosource_write_data: no progress")
    L.smile_extaudiosource_set_external_eoi(obj, b"extsource")
    th.join(240)
    if res.get("rc", -1) != 0:
        raise SystemExit(f"smile_run: {res.get('rc')}: {L.smile_error_msg(obj)}")
    t_end = time.time()
    L.smile_free(obj)
    out = np.frombuffer(b"".join(rows), np.float32).reshape(len(rows), -1) if rows else np.zeros((0, 0), np.float32)
    np.save(out_path, out)
    print(len(rows), "vectors")
    if timing:
        import json
        print(json.dumps({"vectors": len(rows), "feed_to_end_s": t_end - t0, "init_s": t0 - t_init}))


if __name__ == "__main__":
    main()
