"""Image-codec worker of the input pipeline: a plain child PROCESS (started as a script - it imports numpy and PIL, never torch or the
HIP library, and never opens the GPU) that decodes encoded images into slots of a shared-memory block.

Why processes: ``vision.Decode()`` of the reference runs inside ``num_parallel_workers`` dataset processes
(mindpose/data/data_factory.py:116-151).  PIL's JPEG decoder holds the interpreter lock for most of a decode + the copy out of the
image object, so a THREAD pool tops out at ~2x one core (measured: 2.9 -> 1.4 ms per 640x480 image on eight threads); eight
processes writing into shared memory reach 0.42 ms per image.

Protocol (binary, over the worker's stdin / stdout): request = ``<slot:int64> <nbytes:int64>`` + the encoded payload (nbytes < 0:
the payload is a file PATH of -nbytes bytes, utf-8 - the worker reads the file); reply =
``<slot:int64> <height:int64> <width:int64>`` - height -1: the decoded image does not fit a slot (the parent decodes it itself),
-2: the payload could not be decoded.  nbytes 0 ends the worker.  A slot is ``slot_bytes`` of the block, image = uint8 [H, W, 3]
from the slot's first byte.
"""
import io
import struct
import sys


def _decode(buf: bytes):
    import numpy as np
    if buf[:6] == b"\x93NUMPY":
        return np.load(io.BytesIO(buf), allow_pickle=False)
    from PIL import Image
    with Image.open(io.BytesIO(buf)) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        else:
            im.load()
        return np.asarray(im)


def main() -> int:
    import numpy as np
    from multiprocessing import resource_tracker, shared_memory
    name, slot_bytes = sys.argv[1], int(sys.argv[2])
    shm = shared_memory.SharedMemory(name=name)
    try:  # the PARENT owns the block: this process must not unlink it at exit (Python < 3.13 registers every attach)
        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:  # noqa: BLE001
        pass
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    try:
        while True:
            head = inp.read(16)
            if len(head) < 16:
                break
            slot, nbytes = struct.unpack("<qq", head)
            if nbytes == 0:
                break
            payload = inp.read(abs(nbytes))
            if len(payload) < abs(nbytes):
                break
            h = w = -2
            try:
                if nbytes < 0:
                    with open(payload.decode("utf-8"), "rb") as f:
                        payload = f.read()
                arr = _decode(payload)
                if arr.ndim == 3 and arr.shape[2] == 3 and arr.dtype == np.uint8:
                    if arr.nbytes <= slot_bytes:
                        dst = np.ndarray(arr.shape, np.uint8, buffer=shm.buf, offset=slot * slot_bytes)
                        np.copyto(dst, arr)
                        h, w = arr.shape[0], arr.shape[1]
                    else:
                        h = w = -1
            except Exception:  # noqa: BLE001 - reported to the parent, which raises with the sample's context
                h = w = -2
            out.write(struct.pack("<qqq", slot, h, w))
            out.flush()
    finally:
        shm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
