#!/usr/bin/env python3
# SPDX-License-Identifier: GPL-3.0-or-later
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (needs oracle/_ref/libmmref.so, i.e. the
reference compiled from /root/reference by oracle/Makefile).  The fixtures are
data: inputs plus the outputs the reference produced for them.

  kat_matcher.json   known-answer vectors held by the reference's own
                     tests/test_monkey_moore.cpp (inputs + asserted outputs),
                     re-checked here against the compiled reference
  kat_engine.json    the same for tests/test_search_engine.cpp
  diff_search.json   differential vectors: random / adversarial inputs ->
                     MonkeyMoore<T>::search output of the reference
  diff_engine.json   the same for SearchEngine<T>::run (block sizes, endianness)

Usage: python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from _oracle import Ref  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "tests", "golden")

HIRAGANA = "あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほまみむめもやゆよらりるれろわをゃっゅょ"
UNICODE_HIRAGANA = "ぁあぃいぅうぇえぉおかがきぎくぐけげこごさざしじすずせぜそぞただちぢっつづてでとどなにぬねのはばぱひびぴふぶぷへべぺほぼぽまみむめもゃやゅゆょよらりるれろゎわゐゑをんゔゕゖ゙゚゛゜ゝゞゟ"


def shift_alpha(seq, lower, upper, bits):
    """tests/common.hpp:113-128 (shift_alpha_values)."""
    out = []
    mask = (1 << bits) - 1
    for v in seq:
        if 97 <= v <= 122:
            v = (v + lower) & mask
        elif 65 <= v <= 90:
            v = (v + upper) & mask
        out.append(v)
    return out


def chars(s):
    return [ord(c) for c in s]


def kat_matcher():
    """Inputs and asserted outputs of tests/test_monkey_moore.cpp."""
    k = []

    def add(name, elem, data, expect, keyword=None, wildcard=0, seq=None, values=None, maps=None, cite=""):
        k.append(dict(name=name, elem_bytes=elem, data=list(map(int, data)), keyword=keyword, wildcard=wildcard,
                      char_seq=seq, values=values, expect=expect, maps=maps, cite=cite))

    d = shift_alpha(chars("dddccacatchaat"), 3, 3, 8)
    add("u8 ascii catch", 1, d, [6], chars("catch"), maps=[{"a": 97 + 3, "A": 65 + 3}], cite="test_monkey_moore.cpp:16-27")
    add("u8 ascii maca none", 1, d, [], chars("maca"), cite=":29-35")
    seq = chars("aiueobcdfghjklmnpqrstvwxyz")
    d2 = chars("auqqtkcaoaugka")
    add("u8 custom seq match", 1, d2, [8], chars("match"), 0, seq,
        maps=[{chr(c): ord("a") + i for i, c in enumerate(seq)}], cite=":38-52")
    d3 = chars("question of price") + [0] + chars("the last wish") + [0]
    d3 = shift_alpha(d3, -16, -16, 16)
    add("u16 ascii price", 2, d3, [12], chars("price"), maps=[{"a": 97 - 16, "A": 65 - 16}], cite=":57-70")
    add("u16 ascii station none", 2, d3, [], chars("station"), cite=":72-78")
    hseq = chars(HIRAGANA)
    d4 = [1, 12, 16, 110, 44, 16, 12, 16, 17, 26, 110, 22, 44, 22, 110, 26, 21, 45, 110, 31, 7, 31, 13]
    add("u16 hiragana", 2, d4, [4], chars("わたしたちは"), 0, hseq,
        maps=[{chr(c): 1 + i for i, c in enumerate(hseq)}], cite=":81-104")
    d5 = shift_alpha(chars("thebittertasteoflemonwithbutter,"), 8, 8, 8)
    add("u8 wc b*tter", 1, d5, [3, 25], chars("b*tter"), ord("*"), maps=[{"a": 105, "A": 73}] * 2, cite=":112-127")
    add("u8 wc t?ste", 1, d5, [9], chars("t?ste"), ord("?"), maps=[{"a": 105, "A": 73}], cite=":129-137")
    add("u8 wc past* none (wildcard 0)", 1, d5, [], chars("past*"), 0, cite=":139-145")
    d6 = shift_alpha(chars("TheBitterTruthAboutBetterButter."), -32, 24, 8)
    add("u8 mixed B*tter", 1, d6, [3, 19, 25], chars("B*tter"), ord("*"),
        maps=[{"a": 97 - 32, "A": 65 + 24}] * 3, cite=":148-165")
    add("u8 mixed Matter none", 1, d6, [], chars("Matter"), 0, cite=":167-173")
    add("u8 custom seq *at*h", 1, d2, [8], chars("*at*h"), ord("*"), seq,
        maps=[{chr(c): ord("a") + i for i, c in enumerate(seq)}], cite=":177-191")
    d7 = shift_alpha(chars("They muttered: Butter, BETTER, Butcher, matter"), 15, -9, 16)
    add("u16 But**er", 2, d7, [31], chars("But**er"), ord("*"), maps=[{"a": 97 + 15, "A": 65 - 9}], cite=":196-212")
    add("u16 *ITTER none", 2, d7, [], chars("*ITTER"), ord("*"), cite=":214-220")
    kseq = chars(HIRAGANA + "学校行")
    d8 = [1, 12, 16, 26, 111, 50, 51, 22, 111, 52, 7, 31, 13, 6, 112, 111, 44, 16, 12, 35, 111, 52, 7, 16, 2, 113]
    add("u16 kanji seq", 2, d8, [5], chars("**に*行きますか"), ord("*"), kseq,
        maps=[{chr(c): 1 + i for i, c in enumerate(kseq)}], cite=":223-246")
    d9 = [0x00, 0x00, 0x25, 0x26, 0x25, 0x26, 0x27, 0x28, 0x29, 0x30, 0x20, 0x20, 0x00, 0x00, 0x01, 0x00,
          0x01, 0x00, 0x00, 0x89, 0x00, 0x76, 0x77, 0x78, 0x79, 0x7A, 0x81, 0x00, 0x00, 0x01, 0x00, 0x00]
    add("u8 value scan", 1, d9, [4, 21], values=[60, 61, 62, 63, 64, 71], maps=[{}, {}], cite=":252-265")
    add("u8 value scan none", 1, d9, [], values=[80, 81, 82, 83, 84, 85, 86], cite=":267-273")
    d10 = [0x0000, 0x0100, 0x0135, 0x0136, 0x0135, 0x0136, 0x0137, 0x0138, 0x0139, 0x0140, 0x0120, 0x0120, 0x0000,
           0x0100, 0x0101, 0x0000, 0x0101, 0x0089, 0x0000, 0x0045, 0x0046, 0x0047, 0x0048, 0x0049, 0x0050, 0x0000,
           0x0100, 0x0000, 0x0100, 0x0001, 0x0100, 0x0000]
    add("u16 value scan", 2, d10, [4, 19], values=[105, 106, 107, 108, 109, 116], maps=[{}, {}], cite=":277-292")
    add("u16 value scan none", 2, d10, [], values=[200, 201, 205, 208, 209], cite=":294-300")
    d11 = [0x98, 0x94, 0x00, 0xFF, 0xFF, 0x00, 0x01, 0xA5, 0xA1, 0x94, 0x85, 0x98, 0x94]
    add("u8 skip table 0xFF", 1, d11, [9], chars("text"), cite=":315-328")
    d12 = [0x1098, 0x1094, 0x0000, 0xFFFF, 0xFFFF, 0x1000, 0x1001, 0x10A5, 0x10A1, 0x1094, 0x1085, 0x1098, 0x1094]
    add("u16 skip table 0xFFFF", 2, d12, [9], chars("text"), cite=":330-343")
    return k


ENGINE_U8 = [
    0x94, 0x85, 0x98, 0x94, 0x10, 0x10, 0x11, 0x11, 0x00, 0x94, 0x85, 0x98, 0x94, 0x00, 0xFF, 0xFF,
    0x00, 0x00, 0x01, 0x0A, 0xFF, 0xFF, 0x00, 0x00, 0x00, 0x94, 0x85, 0x94, 0x85, 0x98, 0x94, 0x00,
    0xFF, 0x00, 0x0A, 0xFF, 0xFF, 0x01, 0x00, 0x00, 0xFF, 0x00, 0x0A, 0xFF, 0xFF, 0x01, 0x00, 0x00,
    0x00, 0xFF, 0x94, 0x85, 0x98, 0x94, 0x00, 0xFF, 0x00, 0x01, 0xA5, 0xA1, 0x94, 0x85, 0x98, 0x94,
]
ENGINE_U16 = [
    0x1094, 0x1085, 0x1098, 0x1094, 0x0010, 0x0010, 0x0011, 0x0011, 0x0000, 0x1094, 0x1085, 0x1098, 0x1094, 0x0000,
    0xFFFF, 0xFFFF, 0x0000, 0x0000, 0x0001, 0x000A, 0xFFFF, 0xFFFF, 0x0000, 0x0000, 0x0000, 0x1094, 0x1085, 0x1094,
    0x1085, 0x1098, 0x1094, 0x0000, 0xFFFF, 0x0000, 0x000A, 0xFFFF, 0xFFFF, 0x0001, 0x0000, 0x0000, 0xFFFF, 0x0000,
    0x000A, 0xFFFF, 0xFFFF, 0x0001, 0x0000, 0x0000, 0x0000, 0xFFFF, 0x1094, 0x1085, 0x1098, 0x1094, 0x0000, 0x00FF,
    0x0000, 0x0110, 0xA510, 0x01A1, 0x1094, 0x1085, 0x1098, 0x1094,
]


def text_file(text, offset, elem):
    """tests/common.hpp:33-45 (TempFile(text, offset)): each char + offset, as DataType."""
    mask = 0xFF if elem == 1 else 0xFFFF
    vals = [(ord(c) + offset) & mask for c in text]
    a = np.array(vals, dtype=np.uint8 if elem == 1 else np.uint16)
    return a.view(np.uint8).tolist()


def kat_engine():
    """Inputs and asserted outputs of tests/test_search_engine.cpp."""
    k = []

    def add(name, elem, file_bytes, expect, keyword, block_sizes, threads=(1,), wildcard=ord("*"), seq=None,
            big_endian=False, preview_width=50, previews=None, expect_count=None, cite=""):
        k.append(dict(name=name, elem_bytes=elem, file=list(map(int, file_bytes)), keyword=keyword, wildcard=wildcard,
                      char_seq=seq, big_endian=big_endian, block_sizes=list(block_sizes), threads=list(threads),
                      preview_width=preview_width, expect=expect, previews=previews, expect_count=expect_count, cite=cite))

    add("u8 text", 1, ENGINE_U8, [0, 9, 27, 50, 60], chars("text"), [128, 8, 23, 29], (1, 4), preview_width=4,
        cite="test_search_engine.cpp:26-81")
    le = np.array(ENGINE_U16, dtype="<u2").view(np.uint8).tolist()
    be = np.array(ENGINE_U16, dtype=">u2").view(np.uint8).tolist()
    add("u16 LE text", 2, le, [0, 18, 54, 100, 120], chars("text"), [256, 16, 47, 58], (1, 4), cite=":83-137")
    add("u16 BE text", 2, be, [0, 18, 54, 100, 120], chars("text"), [512, 24, 47, 58], (1, 4), big_endian=True, cite=":139-158")
    theater = "#####the theater's theatrical theatergoer thanked the theatrical theater's theatrics####"
    add("u8 preview theater", 1, text_file(theater, 0x10, 1), [9, 30, 65], chars("theater"), [16], preview_width=25,
        previews=["#####the#theater#s#theatr", "eatrical#theatergoer#than", "eatrical#theater#s#theatr"], cite=":168-184")
    add("u8 preview start", 1, text_file("match me please# ", 0x0A, 1), [0], chars("match"), [16], preview_width=8,
        previews=["match#me"], cite=":186-201")
    add("u8 preview end", 1, text_file("###reach the final", 0x2A, 1), [13], chars("final"), [16], preview_width=9,
        previews=["the#final"], cite=":203-218")
    add("u8 preview long", 1, text_file("community#understanding#information", -0x1F, 1), [10], chars("understanding"), [16],
        preview_width=11, previews=["nderstandin"], cite=":220-235")
    add("u16 preview theater", 2, text_file(theater, 0x20, 2), [18, 60, 130], chars("theater"), [32], preview_width=25,
        previews=["#####the#theater#s#theatr", "eatrical#theatergoer#than", "eatrical#theater#s#theatr"], cite=":245-261")
    add("u16 preview start", 2, text_file("catch me please# ", 0, 2), [0], chars("catch"), [32], preview_width=8,
        previews=["catch#me"], cite=":263-278")
    add("u16 preview end", 2, text_file("###the final step", 0, 2), [26], chars("step"), [32], preview_width=9,
        previews=["inal#step"], cite=":280-295")
    content = "あした、わたしたちは、にわに、はなを、まきます"
    u8 = [((ord(c) - 0x3000) & 0xFF) for c in content]      # tests/common.hpp:130-139
    add("u8 hiragana preview", 1, u8, [4], chars("わたしたちは"), [64], seq=chars(UNICODE_HIRAGANA), preview_width=14,
        previews=["あした#わたしたちは#にわに"], cite=":307-326")
    u16 = np.array([ord(c) for c in content], dtype="<u2").view(np.uint8).tolist()
    add("u16 hiragana preview", 2, u16, [8], chars("わたしたちは"), [64], seq=chars(UNICODE_HIRAGANA), preview_width=14,
        previews=["あした#わたしたちは#にわに"], cite=":328-347")
    atch = "match#catch#batch#match#patch#hatch#match"
    add("u8 custom wildcard $atch", 1, text_file(atch, -0x15, 1), None, chars("$atch"), [20], wildcard=ord("$"),
        expect_count=7, cite=":429-447")
    return k


def diff_vectors(ref, rng, n_cases):
    """Random + adversarial inputs through the reference matcher."""
    out = []
    styles = ["uniform", "alphabet", "constant", "ramp", "period"]
    for t in range(n_cases):
        elem = 1 if rng.random() < 0.6 else 2
        hi = 256 if elem == 1 else 65536
        n = int(rng.integers(0, 220))
        L = int(rng.integers(2, 17))
        kwmode = int(rng.integers(0, 5))
        if kwmode == 0:
            kw = [int(rng.integers(97, 123)) for _ in range(L)]
        elif kwmode == 1:
            b = int(rng.integers(97, 120))
            kw = [b + int(rng.integers(0, 3)) for _ in range(L)]
        elif kwmode == 2:
            kw = [int(rng.integers(97, 123)) if rng.random() < 0.7 else int(rng.integers(65, 91)) for _ in range(L)]
        elif kwmode == 3:
            kw = [97] * L
        else:
            kw = [int(rng.integers(33, 127)) for _ in range(L)]
        wildcard = 0
        if rng.random() < 0.45:
            wildcard = ord("*")
            kw = [wildcard if rng.random() < 0.25 else c for c in kw]
        values = None
        if rng.random() < 0.12:
            values = [int(rng.integers(0, 220)) for _ in range(L)]
        style = styles[int(rng.integers(0, len(styles)))]
        if style == "uniform":
            d = rng.integers(0, hi, n)
        elif style == "alphabet":
            k = int(rng.integers(2, 8))
            d = rng.integers(0, k, n) + int(rng.integers(0, hi - k))
        elif style == "constant":
            d = np.full(n, int(rng.integers(0, 2)) * (hi - 1))
        elif style == "ramp":
            d = (np.arange(n) * int(rng.integers(1, 3)) + int(rng.integers(0, hi))) % hi
        else:
            per = int(rng.integers(2, 5))
            d = (rng.integers(0, hi, per)[np.arange(n) % per])
        d = d.astype(np.int64)
        base = values if values is not None else kw
        if n > L + 1:
            for _ in range(int(rng.integers(0, 5))):
                pos = int(rng.integers(0, n - L))
                sh = int(rng.integers(-60, 60))
                for j, v in enumerate(base):
                    if values is None and v == wildcard:
                        continue
                    d[pos + j] = (v + sh) % hi
        d = d.astype(np.uint8 if elem == 1 else np.uint16)
        try:
            if values is not None:
                res = ref.value_scan(elem, values, d)
            else:
                res = ref.search(elem, kw, d, wildcard)
        except RuntimeError:
            continue
        if values is None:
            lead = 0
            while lead < L and kw[lead] == wildcard and wildcard != 0:
                lead += 1
            # keywords the reference cannot terminate on are never generated as goldens
        out.append(dict(elem_bytes=elem, keyword=None if values is not None else kw, wildcard=wildcard, values=values,
                        data=d.tolist(), expect=[int(x) for x in res]))
    return out


def safe_kw(kw, wildcard):
    """Keywords on which the reference terminates."""
    L = len(kw)
    if L < 2:
        return False
    uppers = sum(1 for c in kw if 65 <= c <= 90)
    lowers = sum(1 for c in kw if 97 <= c <= 122)
    norm = list(kw)
    if uppers and lowers:
        if uppers > lowers:
            norm = [wildcard if 97 <= c <= 122 else c for c in norm]
        else:
            norm = [wildcard if 65 <= c <= 90 else c for c in norm]
    is_wc = (wildcard in kw) or (uppers and lowers)
    if not is_wc:
        return True
    lead = 0
    while lead < L and norm[lead] == wildcard:
        lead += 1
    return L - 1 - lead >= 1


def main():
    ref = Ref()
    os.makedirs(GOLDEN, exist_ok=True)

    km = kat_matcher()
    for c in km:
        d = np.array(c["data"], dtype=np.uint8 if c["elem_bytes"] == 1 else np.uint16)
        if c["values"] is not None:
            got = ref.value_scan(c["elem_bytes"], c["values"], d)
        else:
            got = ref.search(c["elem_bytes"], c["keyword"], d, c["wildcard"], c["char_seq"])
        assert got.tolist() == c["expect"], (c["name"], got, c["expect"])
        if c["maps"]:
            for i, m in enumerate(c["maps"]):
                rm = ref.result_map(i)
                assert rm == {ord(k): v for k, v in m.items()}, (c["name"], rm, m)
    json.dump(km, open(os.path.join(GOLDEN, "kat_matcher.json"), "w"), ensure_ascii=False, indent=0)
    print("kat_matcher:", len(km), "cases, reference agrees")

    ke = kat_engine()
    for c in ke:
        for bs in c["block_sizes"]:
            for th in c["threads"]:
                got = ref.engine(c["elem_bytes"], np.array(c["file"], np.uint8), c["keyword"], c["wildcard"], c["char_seq"],
                                 big_endian=c["big_endian"], threads=th, block_size=bs, preview_width=c["preview_width"],
                                 previews=c["previews"] is not None)
                if c["expect"] is not None:
                    assert got.tolist() == c["expect"], (c["name"], bs, got)
                else:
                    assert len(got) == c["expect_count"], (c["name"], got)
                    c["expect"] = [int(x) for x in got]      # the reference test only pins the count; keep its offsets
                if c["previews"] is not None:
                    pv = [ref.result_preview(i) for i in range(len(got))]
                    assert pv == c["previews"], (c["name"], pv)
    json.dump(ke, open(os.path.join(GOLDEN, "kat_engine.json"), "w"), ensure_ascii=False, indent=0)
    print("kat_engine:", len(ke), "cases, reference agrees")

    rng = np.random.default_rng(20261003)
    raw = diff_vectors_safe(ref, rng, 1800)
    json.dump(raw, open(os.path.join(GOLDEN, "diff_search.json"), "w"), separators=(",", ":"))
    print("diff_search:", len(raw), "cases,", sum(1 for c in raw if c["expect"]), "non-empty")

    eng = diff_engine(ref, rng, 700)
    json.dump(eng, open(os.path.join(GOLDEN, "diff_engine.json"), "w"), separators=(",", ":"))
    print("diff_engine:", len(eng), "cases,", sum(1 for c in eng if c["expect"]), "non-empty")


def diff_vectors_safe(ref, rng, n):
    """diff_vectors restricted to keywords the reference terminates on."""
    out = []
    orig_search = ref.search

    def guarded(elem, kw, d, wildcard=0, seq=None):
        if not safe_kw(kw, wildcard):
            raise RuntimeError("unsafe keyword")
        return orig_search(elem, kw, d, wildcard, seq)

    ref.search = guarded
    try:
        out = diff_vectors(ref, rng, n)
    finally:
        ref.search = orig_search
    return out


def diff_engine(ref, rng, n_cases):
    out = []
    for t in range(n_cases):
        elem = 1 if rng.random() < 0.5 else 2
        hi = 256 if elem == 1 else 65536
        nbytes = int(rng.integers(1, 420))
        L = int(rng.integers(3, 13))
        kw = [int(rng.integers(97, 123)) for _ in range(L)]
        wildcard = ord("*")
        if rng.random() < 0.4:
            kw = [wildcard if (0 < i and rng.random() < 0.25) else c for i, c in enumerate(kw)]
        if not safe_kw(kw, wildcard):
            continue
        style = int(rng.integers(0, 4))
        nel = nbytes // elem + 2
        if style == 0:
            d = rng.integers(0, hi, nel)
        elif style == 1:
            k = int(rng.integers(2, 6))
            d = rng.integers(0, k, nel) + int(rng.integers(0, hi - k))
        elif style == 2:
            d = np.full(nel, int(rng.integers(0, hi)))
        else:
            d = (np.arange(nel) + int(rng.integers(0, hi))) % hi
        be = bool(rng.integers(0, 2)) if elem == 2 else False
        arr = d.astype(np.uint8 if elem == 1 else (">u2" if be else "<u2"))
        fb = bytearray(arr.view(np.uint8).tobytes()[:nbytes])
        # plant matches at arbitrary BYTE offsets (odd offsets too for 16 bit)
        for _ in range(int(rng.integers(0, 6))):
            if nbytes <= L * elem + 1:
                break
            pos = int(rng.integers(0, nbytes - L * elem))
            sh = int(rng.integers(0, 60))
            for j, v in enumerate(kw):
                if v == wildcard:
                    continue
                val = (v + sh) % hi
                if elem == 1:
                    fb[pos + j] = val
                else:
                    b = val.to_bytes(2, "big" if be else "little")
                    fb[pos + 2 * j] = b[0]
                    fb[pos + 2 * j + 1] = b[1]
        block = int(rng.integers(2 if elem == 1 else 4, 200))
        fbn = np.frombuffer(bytes(fb), dtype=np.uint8)
        res = ref.engine(elem, fbn, kw, wildcard, None, big_endian=be, threads=int(rng.integers(1, 4)), block_size=block)
        out.append(dict(elem_bytes=elem, keyword=kw, wildcard=wildcard, big_endian=be, block_size=block,
                        file=fbn.tolist(), expect=[int(x) for x in res]))
    return out


if __name__ == "__main__":
    main()
