"""CPU tier: the reference-side binding of INTEGRATION.md is code that COMPILES against the reference's own headers and LINKS against
libt2gpu.so. integration/*_gpu.cpp are the bodies of the reference's slots as calls into the C ABI; `make -C oracle
binding` compiles them with the reference's flags, the image's Qt 5.9.7 headers and the reference's headers where they lie, and links
them with the reference's own objects (oracle/_ref/libref_t2rx_gpu*.so). Skipped where /root/reference or the Qt SDK is absent (the GPU
box: tests/test_binding_gpu.py RUNS the prebuilt libraries there). Nothing of the reference is committed."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/DVB_T2"
have_ref = os.path.isdir(REF) and os.path.exists("/opt/conda/bin/moc")
needs_ref = pytest.mark.skipif(not have_ref, reason="/root/reference or the Qt SDK (moc) is not here")
SLOTS = {   # file -> (reference header that declares the slot, its line there, mangled-name pattern of the slot's definition)
    "dvbt2_demodulator_gpu.cpp": ("dvbt2_demodulator.h", 78, r"_ZN17dvbt2_demodulator7executeE"),
    "time_deinterleaver_gpu.cpp": ("time_deinterleaver.h", 45, r"_ZN18time_deinterleaver(7execute|14l1_dyn_execute)E"),
    "llr_demapper_gpu.cpp": ("llr_demapper.h", 44, r"_ZN12llr_demapper7executeE"),
    "ldpc_decoder_gpu.cpp": ("ldpc_decoder.h", 90, r"_ZN12ldpc_decoder7executeE"),
    "bch_decoder_gpu.cpp": ("bch_decoder.h", 41, r"_ZN11bch_decoder7executeE"),
    "bb_de_header_gpu.cpp": ("bb_de_header.h", 59, r"_ZN12bb_de_header7executeE"),
}


def nm(path, *flags):
    return subprocess.run(["nm"] + list(flags) + [path], stdout=subprocess.PIPE, text=True, check=True).stdout


def test_integration_md_quotes_the_binding_files_verbatim():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration_md.py"), "--check"]).returncode == 0, \
        "INTEGRATION.md is stale: run python tools/gen_integration_md.py"
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for fn in SLOTS:
        assert "<!-- BEGIN integration/%s -->" % fn in text, fn


@needs_ref
def test_slot_declarations_are_where_the_binding_says(built):
    """the reference header line each file cites does declare the slot it replaces"""
    for fn, (hdr, line, _) in SLOTS.items():
        src = open(os.path.join(REF, hdr)).read().splitlines()
        assert "execute(" in src[line - 1], (hdr, line, src[line - 1])
        assert "%s:%d" % (hdr, line) in open(os.path.join(ROOT, "integration", fn)).read().replace("-%d" % line, ":%d" % line) or \
            ("%s:" % hdr) in open(os.path.join(ROOT, "integration", fn)).read(), fn


@needs_ref
def test_binding_compiles_against_the_reference_headers_and_links(built):
    from test_capi_symbols import declared_functions
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "binding"])
    bind = os.path.join(ROOT, "oracle", "_ref", "bind")
    used = set()
    for fn, (_, _, pattern) in SLOTS.items():
        obj = os.path.join(bind, fn.replace(".cpp", ".o"))
        syms = nm(obj)
        assert re.search(r" T " + pattern, syms), "%s does not define its slot" % fn          # compiled: syntax and types against the real class
        used |= set(re.findall(r" U (t2gpu_\w+)", syms))
        ref_obj = os.path.join(bind, fn.replace("_gpu.cpp", ".ref.o"))
        assert re.search(r" t " + pattern, nm(ref_obj)) and not re.search(r" T " + pattern, nm(ref_obj)), "reference definition of %s not localized" % fn
    assert len(used) >= 15 and used <= set(declared_functions()), used - set(declared_functions())
    exported = set(re.findall(r" T (t2gpu_\w+)", nm(os.path.join(ROOT, "sdr_receiver_dvb_t2_amd", "libt2gpu.so"), "-D")))
    assert used <= exported, used - exported
    for lib, n_slots in (("libref_t2rx_gpufec.so", 3), ("libref_t2rx_gpu.so", 6)):
        path = os.path.join(ROOT, "oracle", "_ref", lib)
        dyn = nm(path, "-D")
        assert len(set(re.findall(r" U (t2gpu_\w+)", dyn))) >= 8, lib                        # linked against the library, symbols left to it
        assert "libt2gpu.so" in subprocess.run(["readelf", "-d", path], stdout=subprocess.PIPE, text=True).stdout


@needs_ref
def test_public_header_is_qt_safe(tmp_path):
    """include/t2gpu.h after Qt's headers (which define `signals`, `slots`, `emit`, `foreach` as macros): found broken by the binding"""
    src = tmp_path / "qt_then_t2gpu.cpp"
    src.write_text('#include <QtCore/QObject>\n#include "t2gpu.h"\n#include "t2gpu_stages.hpp"\nint main() { return t2gpu_version() > 0 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-fsyntax-only", "-I/opt/conda/include/qt", "-I/opt/conda/include/qt/QtCore",
                           "-I" + os.path.join(ROOT, "include"), str(src)])
