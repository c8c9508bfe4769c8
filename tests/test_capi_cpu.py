"""CPU-side tests of the drop-in boundary: libneedle_capi.so loads, exports every symbol the headers
declare, reproduces the reference's constructor/argument behaviour (needle-capi/src/lib.rs), and its
host-only pieces (FrameHashes files, header MD5, file discovery) agree with the oracle.  No compute entry
point is called with a GPU here; where one is called it must fail loudly because there is no CPU path."""
import ctypes as C
import os
import re
import shutil
import struct
import subprocess

import numpy as np
import pytest

from needle_amd import capi, synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(needle_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for header, listed in [("needle.h", capi.NEEDLE_H_SYMBOLS), ("needle_hip.h", capi.NEEDLE_HIP_H_SYMBOLS)]:
        declared = _declared(header)
        assert declared == sorted(listed), f"{header} and capi.py disagree"
        for sym in declared:
            assert hasattr(L, sym), f"{sym} declared in {header} but not exported"
    assert len(capi.NEEDLE_H_SYMBOLS) == 13   # needle-capi/needle.h:146-248


def test_error_enum_and_strings():
    """repr(C) values 0..11 and the exact strings of needle_error_to_str (lib.rs:58-85,138-199)."""
    want = ["No error", "Invalid UTF-8 string", "Input argument is NULL",
            "One or more input arguments were invalid (usually zero)", "Frame hash data not found on disk",
            "Frame hash data has an invalid version.", "Invalid frame hash data read from disk",
            "Comparator requires at least 2 video paths", "Analyzer hash period must be greater than 0",
            "Analyzer hash duration must be greater than 3 seconds", "I/O error",
            "Unknown error occurred; please re-run with logging enabled"]
    assert [capi.error_to_str(i) for i in range(12)] == want
    header = open(os.path.join(ROOT, "include", "needle.h")).read()
    names = re.findall(r"NeedleError_(\w+)", header.split("typedef enum NeedleError")[1].split("} NeedleError;")[0])
    assert names == capi.ERROR_NAMES


def _cpaths(paths):
    arr = (C.c_char_p * len(paths))(*[p if isinstance(p, bytes) else p.encode() for p in paths])
    return C.cast(arr, C.POINTER(C.c_char_p)), arr


def test_analyzer_constructors_like_reference_tests():
    """needle-capi/src/lib.rs:681-703: _new_default on a path that need not exist -> Ok, non-null; free."""
    L = capi.lib()
    ptr, keep = _cpaths(["/tmp/abcd.mkv"])
    out = C.c_void_p()
    assert L.needle_audio_analyzer_new_default(ptr, 1, C.byref(out)) == 0 and out.value
    fh = C.c_void_p()
    assert L.needle_audio_analyzer_get_frame_hashes(out, 0, C.byref(fh)) == 3     # nothing run yet: InvalidArgument
    assert L.needle_audio_analyzer_get_frame_hashes(None, 0, C.byref(fh)) == 2
    assert L.needle_audio_analyzer_run(out, 0.0, False, True) == 9               # hash_duration <= 0 (lib.rs:474)
    assert L.needle_audio_analyzer_run(None, 0.3, False, True) == 2
    L.needle_audio_analyzer_free(out)
    L.needle_audio_analyzer_free(None)
    assert L.needle_audio_analyzer_new_default(None, 1, C.byref(out)) == 2        # NullArgument (lib.rs:383)
    assert L.needle_audio_analyzer_new_default(ptr, 1, None) == 2
    bad, keep2 = _cpaths([b"/tmp/\xff\xfe.mkv"])
    assert L.needle_audio_analyzer_new_default(bad, 1, C.byref(out)) == 1         # InvalidUtf8String (lib.rs:297)
    nullelem = (C.c_char_p * 1)(None)
    assert L.needle_audio_analyzer_new_default(C.cast(nullelem, C.POINTER(C.c_char_p)), 1, C.byref(out)) == 2


def test_comparator_constructors_like_reference_tests():
    """lib.rs:705-739 plus the minimum-paths rule (lib.rs:569-571)."""
    L = capi.lib()
    ptr, keep = _cpaths(["/tmp/abcd.mkv", "/tmp/efgh.mp4"])
    out = C.c_void_p()
    assert L.needle_audio_comparator_new(ptr, 2, False, 10, 10, 10, 0.0, C.byref(out)) == 0 and out.value
    L.needle_audio_comparator_free(out)
    out = C.c_void_p()
    assert L.needle_audio_comparator_new_default(ptr, 2, C.byref(out)) == 0 and out.value
    assert L.needle_audio_comparator_run(None, True, True, False, False, True) == 2
    # search without analyze and without .needle.dat files: FrameHashDataNotFound (data.rs:106-108)
    assert L.needle_audio_comparator_run(out, False, False, False, False, True) == 4
    L.needle_audio_comparator_free(out)
    L.needle_audio_comparator_free(None)
    assert L.needle_audio_comparator_new_default(ptr, 1, C.byref(out)) == 7       # ComparatorMinimumPaths
    assert L.needle_audio_comparator_new_default(None, 2, C.byref(out)) == 2


def test_print_paths(capfd):
    L = capi.lib()
    ptr, keep = _cpaths(["/tmp/a b.mkv", "/tmp/é.mp4"])
    out = C.c_void_p()
    assert L.needle_audio_analyzer_new_default(ptr, 2, C.byref(out)) == 0
    L.needle_audio_analyzer_print_paths(out)
    L.needle_audio_analyzer_free(out)
    assert capfd.readouterr().out == "/tmp/a b.mkv\n/tmp/é.mp4\n"


def test_find_video_files(tmp_path):
    """lib.rs:208-281 / util.rs:60-96: existing paths only, one directory level, *.needle.dat excluded.
    Decodable media in this build = RIFF/WAVE (FFmpeg is out of scope)."""
    L = capi.lib()
    e = synth.make_episode(0, 2.0, 0.0)
    d = tmp_path / "show"
    d.mkdir()
    synth.write_wav(str(d / "b.wav"), e.pcm)
    synth.write_wav(str(d / "a.wav"), e.pcm)
    (d / "notes.txt").write_text("not media")
    (d / "a.needle.dat").write_bytes(b"RIFF0000WAVE")
    (d / "sub").mkdir()
    synth.write_wav(str(d / "sub" / "deep.wav"), e.pcm)
    single = tmp_path / "single.wav"
    synth.write_wav(str(single), e.pcm, channels=2)
    ptr, keep = _cpaths([str(d), str(single)])
    videos = C.POINTER(C.c_char_p)()
    n = C.c_size_t(0)
    for full in (False, True):
        assert L.needle_util_find_video_files(ptr, 2, full, True, C.byref(videos), C.byref(n)) == 0
        got = [videos[i].decode() for i in range(n.value)]
        assert got == [str(d / "a.wav"), str(d / "b.wav"), str(single)]
        L.needle_util_video_files_free(videos, n)
    L.needle_util_video_files_free(None, 0)
    # `full` looks at the WAV header only: an encoding this build cannot decode (A-law) is dropped, and a
    # 3 GB (sparse) file costs a few header reads, not a pass over its samples
    alaw = bytearray(open(str(single), "rb").read()[:4096])
    alaw[20:22] = (6).to_bytes(2, "little")
    (d / "c-alaw.wav").write_bytes(bytes(alaw))
    big = d / "d-big.wav"
    hdr = bytearray(open(str(single), "rb").read()[:44])
    hdr[40:44] = (0xFFFFFFFF).to_bytes(4, "little")
    big.write_bytes(bytes(hdr))
    os.truncate(str(big), 3 << 30)
    import time
    for full, want in ((False, ["a.wav", "b.wav", "c-alaw.wav", "d-big.wav"]), (True, ["a.wav", "b.wav", "d-big.wav"])):
        t0 = time.perf_counter()
        assert L.needle_util_find_video_files(ptr, 1, full, True, C.byref(videos), C.byref(n)) == 0
        assert time.perf_counter() - t0 < 1.0
        assert [videos[i].decode() for i in range(n.value)] == [str(d / w) for w in want]
        L.needle_util_video_files_free(videos, n)
    assert L.needle_util_find_video_files(ptr, 0, False, True, C.byref(videos), C.byref(n)) == 3
    assert L.needle_util_find_video_files(None, 1, False, True, C.byref(videos), C.byref(n)) == 2
    missing, k2 = _cpaths([str(tmp_path / "nope")])
    assert L.needle_util_find_video_files(missing, 1, False, True, C.byref(videos), C.byref(n)) == 11  # PathNotFound -> Unknown


def test_frame_hashes_file_is_bincode_compatible(tmp_path):
    """Files written through the C ABI are byte-identical to the oracle's (= the reference's bincode layout,
    SURVEY.md Appendix B) and each side reads the other's."""
    opening = [(0xDEADBEEF, 2_600_000_000), (7, 2_846_000_007), (0, 3_092_000_014)]
    ending = [(9, 1_082_600_000_000)]
    md5 = "759c6a520c5ce70359fdff38c4be6b98"
    ours = str(tmp_path / "ours.needle.dat")
    theirs = str(tmp_path / "theirs.needle.dat")
    fh = capi.FrameHashes.new(opening, ending, 300_000_012, md5)
    fh.write(ours)
    assert O.frame_hashes_write(theirs, O.FrameHashes(opening, ending, 300_000_012, md5)) == 0
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    back = capi.FrameHashes.from_path(theirs)
    assert list(zip(*[a.tolist() for a in back.opening_data()])) == opening
    assert list(zip(*[a.tolist() for a in back.ending_data()])) == ending
    assert back.hash_duration() == 300_000_012 and back.md5() == md5
    rc, o = O.frame_hashes_read(ours)
    assert rc == 0 and o.opening == opening and o.ending == ending
    # error mapping (lib.rs:126-128)
    with pytest.raises(capi.NeedleError) as e:
        capi.FrameHashes.from_path(str(tmp_path / "missing.needle.dat"))
    assert e.value.name == "FrameHashDataNotFound"
    raw = open(ours, "rb").read()
    for blob in (raw[:-3], raw[:20], struct.pack("<I", 5) + raw[4:], raw[:4] + struct.pack("<I", 1) + raw[8:]):
        open(ours, "wb").write(blob)
        with pytest.raises(capi.NeedleError) as e:
            capi.FrameHashes.from_path(ours)
        assert e.value.name == "InvalidFrameHashData"


def test_header_md5_golden_and_short_file(tmp_path):
    head = open(os.path.join(HERE, "golden", "sample-5s.header8k.bin"), "rb").read()
    p = tmp_path / "v.mp4"
    p.write_bytes(head + b"x" * 100)
    assert capi.header_md5(str(p)) == "759c6a520c5ce70359fdff38c4be6b98"   # reference snapshot :43
    p.write_bytes(head[:100])
    with pytest.raises(capi.NeedleError) as e:
        capi.header_md5(str(p))
    assert e.value.name == "IOError"


def test_fingerprint_constants_match_oracle():
    L = capi.lib()
    assert L.needle_hip_fingerprint_sample_rate() == 11025
    assert L.needle_hip_fingerprint_delay_ms() == O.delay_ms() == 2600
    assert L.needle_hip_fingerprint_item_duration_ms() == O.item_duration_ms() == 123
    for s in [0, 4095, 4096, 30000, 496125, 7938000, 14883750]:
        assert L.needle_hip_fingerprint_num_items(s) == O.num_items(s)
        assert L.needle_hip_fingerprint_num_kept(s, 2) == (O.num_items(s) + 1) // 2


def test_compute_entry_points_fail_loudly_without_a_gpu(has_gpu):
    """There is no CPU fallback: on a box without a HIP device every compute call returns an error."""
    if has_gpu:
        pytest.skip("a GPU is present; covered by the -m gpu parity tests")
    pcm = np.zeros(20000, np.int16)
    with pytest.raises(capi.NeedleError) as e:
        capi.fingerprint([pcm])
    assert e.value.name == "Unknown" and "no HIP device" in str(e.value)
    with pytest.raises(capi.NeedleError):
        capi.hamming_runs([np.arange(10, dtype=np.uint32), np.arange(10, dtype=np.uint32)], [(0, 1, 1)], 10)
    fhs = [capi.FrameHashes.new([(i, 2_600_000_000 + i * 246_000_000) for i in range(100)], [], 300_000_012)
           for _ in range(2)]
    with pytest.raises(capi.NeedleError):
        capi.Comparator(["a.wav", "b.wav"]).run_with_frame_hashes(fhs)
    with pytest.raises(capi.NeedleError):
        capi.Analyzer(["a.wav"]).run_pcm([pcm])
    with pytest.raises(capi.NeedleError):
        capi.set_device(0)


def test_c_program_links_against_header_and_library(tmp_path):
    """An ordinary C consumer of needle.h (same call sequence as the reference's examples, which CI only
    compiles: .github/workflows/test.yml:39-40) builds with -Wall -Werror and runs."""
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-Wall", "-Werror", "-std=c11", os.path.join(HERE, "c_abi", "abi_smoke.c"),
                    "-I", os.path.join(ROOT, "include"), "-L", os.path.dirname(capi.LIB_PATH), "-lneedle_capi",
                    "-Wl,-rpath," + os.path.dirname(capi.LIB_PATH), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "abi smoke ok" in out


@pytest.mark.skipif(not os.path.isdir("/root/reference/needle-capi/examples"), reason="reference tree not present")
def test_reference_examples_compile_unchanged(tmp_path):
    """needle-capi/examples/*.c compile and link, unmodified and in place, against OUR header and library
    (the reference's own ABI gate; nothing is copied into the repo)."""
    for name in ("analyzer", "comparator", "full"):
        src = f"/root/reference/needle-capi/examples/{name}.c"
        subprocess.run(["gcc", src, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-L",
                        os.path.dirname(capi.LIB_PATH), "-lneedle_capi", "-o", str(tmp_path / f"{name}.out")],
                       check=True)


# ---- libchromaprint-compatible entry points (include/needle_chromaprint.h) ---------------------------------------
def _chromaprint_lib():
    L = C.CDLL(os.path.join(os.path.dirname(capi.LIB_PATH), "libneedle_chromaprint.so"))
    L.chromaprint_new.restype = C.c_void_p
    L.chromaprint_new.argtypes = [C.c_int]
    L.chromaprint_get_version.restype = C.c_char_p
    for fn in ("chromaprint_free", "chromaprint_dealloc"):
        getattr(L, fn).argtypes = [C.c_void_p]
        getattr(L, fn).restype = None
    for fn in ("chromaprint_get_sample_rate", "chromaprint_get_item_duration", "chromaprint_get_item_duration_ms",
               "chromaprint_get_delay", "chromaprint_get_delay_ms", "chromaprint_finish", "chromaprint_get_num_channels",
               "chromaprint_get_algorithm", "chromaprint_clear_fingerprint"):
        getattr(L, fn).argtypes = [C.c_void_p]
    L.chromaprint_start.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.chromaprint_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.chromaprint_get_raw_fingerprint.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_int)]
    L.chromaprint_get_raw_fingerprint_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    return L


def test_chromaprint_compat_exports_and_constants(has_gpu):
    """The libchromaprint calls needle makes (analyzer.rs:176,179,218,275,286,288-289,300) exist with
    libchromaprint's names, 1/0 return convention and the constants chromaprint reports for its default algorithm."""
    L = _chromaprint_lib()
    text = open(os.path.join(ROOT, "include", "needle_chromaprint.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for sym in sorted(set(re.findall(r"\b(chromaprint_[a-z_]+)\s*\(", text))):
        assert hasattr(L, sym), sym
    assert L.chromaprint_new(0) is None and L.chromaprint_new(3) is None      # only TEST2 (the default)
    ctx = L.chromaprint_new(1)
    assert ctx
    assert L.chromaprint_get_sample_rate(ctx) == 11025 and L.chromaprint_get_num_channels(ctx) == 1
    assert (L.chromaprint_get_item_duration(ctx), L.chromaprint_get_item_duration_ms(ctx)) == (1365, 123)
    assert (L.chromaprint_get_delay(ctx), L.chromaprint_get_delay_ms(ctx)) == (28666, 2600)
    assert L.chromaprint_get_delay_ms(ctx) == O.delay_ms() and L.chromaprint_get_item_duration_ms(ctx) == O.item_duration_ms()
    pcm = np.zeros(30000, np.int16)
    assert L.chromaprint_feed(ctx, pcm.ctypes.data, 100) == 0                   # not started
    assert L.chromaprint_start(ctx, 44100, 2) == 1                              # other rates: device resampler first
    assert L.chromaprint_start(ctx, 100, 2) == 0
    assert L.chromaprint_start(ctx, 11025, 6) == 0
    assert L.chromaprint_start(ctx, 11025, 2) == 1
    assert L.chromaprint_feed(ctx, pcm.ctypes.data, 101) == 0                   # not a multiple of the channel count
    assert L.chromaprint_feed(ctx, pcm.ctypes.data, 30000) == 1
    n = C.c_int(-1)
    assert L.chromaprint_get_raw_fingerprint_size(ctx, C.byref(n)) == 0         # before finish
    if not has_gpu:
        assert L.chromaprint_finish(ctx) == 0                                   # no device, no CPU fallback
    L.chromaprint_free(ctx)
    L.chromaprint_free(None)
    L.chromaprint_dealloc(None)


# ---- the command-line front-end (needle/src/main.rs) --------------------------------------------------------------
NEEDLE_BIN = os.path.join(os.path.dirname(capi.LIB_PATH), "..", "bin", "needle")


def _cli(*args):
    import subprocess
    return subprocess.run([NEEDLE_BIN, *args], capture_output=True, text=True, timeout=60)


def test_cli_validation_matches_reference_messages(tmp_path):
    """Cli::validate and the search arm's minimum-paths check (main.rs:196-241,306-317): same messages, exit 2."""
    r = _cli("info")
    assert r.returncode == 0 and "needle version:" in r.stdout and "HIP devices:" in r.stdout
    for args, msg in [
        (["analyze", str(tmp_path), "--opening-search-percentage", "1.0"], "opening_search_percentage must be less than 1.0"),
        (["analyze", str(tmp_path), "--ending-search-percentage=1.5"], "ending_search_percentage must be less than 1.0"),
        (["analyze", str(tmp_path), "--hash-duration", "0"], "hash_duration must be greater than 0"),
        (["search", str(tmp_path), "--hash-match-threshold", "33"], "hash_match_threshold cannot be larger than 32"),
        (["search", str(tmp_path)], "need at least 2 valid video files, but only found 1 in provided video paths"),
        (["search", str(tmp_path / "missing")], "path does not exist"),
        (["analyze"], "required arguments were not provided"),
        (["frobnicate"], "wasn't expected"),
        ([], "requires a subcommand"),
        (["search", str(tmp_path), "--min-opening-duration", "70000"], "Invalid value"),
        (["analyze", str(tmp_path), "--mode", "video"], "isn't a valid value"),
    ]:
        r = _cli(*args)
        assert r.returncode == 2, (args, r.stderr)
        # (the C ABI itself logs "needle error: ..." first where the failure comes from the library, lib.rs:124)
        assert "\nerror: " in "\n" + r.stderr and msg in r.stderr, (args, r.stderr)
    # global flags are accepted on either side of the subcommand; discovery ignores non-media files
    (tmp_path / "notes.txt").write_text("x")
    r = _cli("--no-threading", "search", str(tmp_path), "--file-headers-only", "--no-display")
    assert r.returncode == 2 and "need at least 2 valid video files" in r.stderr


def test_rust_ffi_declarations_name_exported_symbols():
    """rust/needle-hip cannot be compiled here (no toolchain); at least every function its ffi.rs declares must
    be a symbol libneedle_capi.so exports and one of the headers declares."""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = open(os.path.join(root, "rust", "needle-hip", "src", "ffi.rs")).read()
    names = re.findall(r"pub fn (needle_\w+)\(", src)
    assert len(names) > 20
    headers = open(os.path.join(root, "include", "needle.h")).read() + open(os.path.join(root, "include", "needle_hip.h")).read()
    L = capi.lib()
    for n in names:
        assert hasattr(L, n), n
        assert re.search(r"\b%s\(" % n, headers), n


def test_every_function_the_headers_declare_is_exported():
    """include/needle.h and include/needle_hip.h against libneedle_capi.so, include/needle_chromaprint.h against
    libneedle_chromaprint.so: parsed from the headers, so a declaration added without a definition fails here."""
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")

    def declared(header, prefix):
        text = open(os.path.join(root, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))

    L = capi.lib()
    names = declared("needle.h", "needle_") + declared("needle_hip.h", "needle_hip_")
    assert len(names) >= 13 + 40
    for n in names:
        assert hasattr(L, n), n
    assert set(capi.NEEDLE_H_SYMBOLS) == set(declared("needle.h", "needle_"))
    assert set(capi.NEEDLE_HIP_H_SYMBOLS) == set(declared("needle_hip.h", "needle_hip_"))
    CL = _chromaprint_lib()
    cnames = declared("needle_chromaprint.h", "chromaprint_")
    assert len(cnames) >= 10
    for n in cnames:
        assert hasattr(CL, n), n


def test_pair_index_map_matches_the_enumeration(tmp_path):
    """comparator.rs:534-545 enumerates the pairs (i, j), i < j, i-major; the host code maps a pair index back to
    (i, j) in closed form (a library has n^2 / 2 pairs and each needs it).  tests/cpp/pair_index_check.cpp compares
    the map with the enumeration itself."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "needle_amd", "lib")
    exe = str(tmp_path / "pair_index_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(root, "tests", "cpp", "pair_index_check.cpp"),
                    "-L" + libdir, "-lneedle_capi", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "pair_at ok" in out.stdout, out.stdout + out.stderr


def test_first_pass_schedule_covers_every_frame_pair_once(tmp_path):
    """needle_amd/csrc/stft32_schedule.h: the first pass's workgroups get smaller towards the end of each XCD's part of the
    timeline.  tests/cpp/schedule_check.cpp runs the kernel's own range arithmetic on the host for launches from 0 to
    5.9 M pairs: every pair exactly once, each XCD front to back, nothing beyond the launch."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "schedule_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tests", "cpp", "schedule_check.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_rust_ffi_signatures_and_struct_layouts_match_the_headers(tmp_path):
    """The first `cargo build` of rust/needle-hip on a machine with a toolchain should be a formality: every extern
    declaration of ffi.rs has the argument and return types of the C prototype (needle-capi/src/lib.rs:346-637 is the
    surface the 13 reference symbols come from), every #[repr(C)] struct the size and field offsets gcc gives the C
    struct, NeedleError the header's variants in order."""
    from tests import rust_ffi_check as R
    assert R.c_type_to_rust("const char *const *paths") == "*const *const c_char"
    assert R.c_type_to_rust("const char *const **videos") == "*mut *const *const c_char"
    assert R.c_type_to_rust("struct NeedleAudioAnalyzer **output") == "*mut *mut NeedleAudioAnalyzer"
    assert R.c_type_to_rust("const struct NeedleAudioComparator **output") == "*mut *const NeedleAudioComparator"
    assert R.c_type_to_rust("uint64_t counts[4]") == "*mut u64" and R.c_type_to_rust("const uint8_t id[128]") == "*const u8"
    protos = R.c_prototypes()
    fns, structs, variants = R.rust_declarations()
    assert len(fns) >= 50 and len(protos) >= len(fns)
    for name, (params, ret) in fns.items():
        assert name in protos, name
        want_params, want_ret = protos[name]
        assert params == want_params, (name, params, want_params)
        assert ret == want_ret, (name, ret, want_ret)
    assert variants == R.header_error_variants() and len(variants) == 12
    sized = {k: v for k, v in structs.items() if v}              # opaque handles have no fields
    assert {"NeedleHipSearchResult", "NeedleHipRun", "NeedleHipCertAudit"} <= set(sized)
    c = R.c_layout(sized, str(tmp_path))
    for name, fields in sized.items():
        offsets, size = R.rust_layout(fields)
        assert c[(name, "size")] == size, (name, size, c[(name, "size")])
        for f, off in offsets:
            assert c[(name, f)] == off, (name, f, off, c[(name, f)])


def test_round4_entry_points_validate_their_arguments_without_a_device():
    """needle_hip_library_job_runs / _job_comm_bytes / _audit and needle_hip_host_threads: NULLs, bad slots and a library
    without PCM are refused with the reference's error codes before anything touches a device."""
    import ctypes as C
    L = capi.lib()
    vp = C.c_void_p
    ERR = {name: code for code, name in enumerate(capi.ERROR_NAMES)}
    L.needle_hip_library_job_runs.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.needle_hip_library_job_comm_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    L.needle_hip_library_audit.argtypes = [vp, vp]
    runs, n = vp(), C.c_size_t(7)
    assert L.needle_hip_library_job_runs(None, 0, C.byref(runs), C.byref(n)) == ERR["NullArgument"]
    lib = capi.Library(3)
    assert L.needle_hip_library_job_runs(lib._h, 0, None, C.byref(n)) == ERR["NullArgument"]
    assert L.needle_hip_library_job_runs(lib._h, 2, C.byref(runs), C.byref(n)) == ERR["InvalidArgument"]
    assert L.needle_hip_library_job_runs(lib._h, 1, C.byref(runs), C.byref(n)) == ERR["Ok"] and n.value == 0   # no job yet
    b = (C.c_uint64 * 4)(9, 9, 9, 9)
    assert L.needle_hip_library_job_comm_bytes(lib._h, 0, b) == ERR["Ok"] and list(b) == [0, 0, 0, 0]
    assert L.needle_hip_library_job_comm_bytes(lib._h, -1, b) == ERR["InvalidArgument"]
    assert L.needle_hip_library_job_comm_bytes(lib._h, 0, None) == ERR["NullArgument"]
    audit = capi.CCertAudit()
    assert L.needle_hip_library_audit(None, C.byref(audit)) == ERR["NullArgument"]
    assert L.needle_hip_library_audit(lib._h, None) == ERR["NullArgument"]
    assert L.needle_hip_library_audit(lib._h, C.byref(audit)) == ERR["InvalidArgument"]        # no resident PCM
    assert 1 <= capi.host_threads() <= max(1, len(os.sched_getaffinity(0)))
