"""Rules of the library's host code that a compiler does not hold (CPU tier: reads the sources only)."""
import glob
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "photon_amd", "csrc")


def code_lines(path):
    for no, line in enumerate(open(path, encoding="utf-8"), 1):
        yield no, line.split("//", 1)[0]


def test_no_host_side_hipmemset():
    """hipMemset returns once its fill kernel is QUEUED on the null stream (tools/ubench/null_stream_memset.hip: 9 us, the
    fill 200 ms away behind a full chip) and a launch on a non-blocking stream is not ordered behind the null stream: a
    scene's work queues were once zeroed that way and, with eight shards side by side on one device, the fill sometimes
    ran after the march had started (groups marched twice).  Zeroing is either hipMemsetAsync on the consumer's stream
    or photon::device_zero (complete on return), or part of a host-to-device copy."""
    offenders = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))):
        for no, code in code_lines(path):
            if re.search(r"\bhipMemset(D8|D16|D32|2D|3D)?\s*\(", code):
                offenders.append(f"{os.path.basename(path)}:{no}: {code.strip()}")
    assert not offenders, "host-asynchronous hipMemset on the null stream:\n" + "\n".join(offenders)


def test_no_host_side_device_to_device_hipmemcpy():
    """The same for hipMemcpy(..., hipMemcpyDeviceToDevice): it returns in 5 us with the copy still queued on the null
    stream, and a kernel launched afterwards on a non-blocking stream READ THE OLD BYTES in tools/ubench/null_stream_memset.hip.
    Device-to-device copies are hipMemcpyAsync on a named stream, with the wait (if any) written out."""
    offenders = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))):
        text = "".join(code for _, code in code_lines(path))
        for m in re.finditer(r"\bhipMemcpy\s*\(([^;]*);", text):
            if "hipMemcpyDeviceToDevice" in m.group(1) or "hipMemcpyDefault" in m.group(1):
                offenders.append(f"{os.path.basename(path)}: hipMemcpy({' '.join(m.group(1).split())[:100]}")
    assert not offenders, "host-asynchronous device-to-device hipMemcpy on the null stream:\n" + "\n".join(offenders)
