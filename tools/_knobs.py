"""Developer tools and the parity tests' child processes: apply MFA_TEST_KNOBS="name=value,..." through the library's test
hook (mfa_test_set_knob, include/mfa.h) -- the library itself reads no such environment variable."""
import os


def apply():
    spec = os.environ.get("MFA_TEST_KNOBS", "")
    if not spec:
        return
    from mini_flash_attention import capi
    lib = capi.load()
    for kv in spec.split(","):
        name, value = kv.split("=")
        rc = lib.mfa_test_set_knob(name.strip().encode(), int(value))
        assert rc == 0, capi.last_error()
