"""A BAM reader for the tests (no samtools in the image): BGZF blocks are gzip members, so `gzip` reads the
stream; the records are decoded per SAMv1 section 4 and rendered back to SAM text."""
import gzip
import struct

SEQ = "=ACMGRSVTWYHKDBN"
OPS = "MIDNSHP=X"


def bgzf_blocks(data: bytes):
    """(block sizes, True when the file ends with the empty end-of-file block)"""
    at, sizes = 0, []
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block"
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        assert data[at + 12:at + 16] == b"BC\x02\x00" and xlen == 6
        bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
        assert bsize <= 0x10000
        sizes.append(bsize)
        at += bsize
    assert at == len(data)
    return sizes, data[-28:] == bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def to_sam(path) -> str:
    raw = open(path, "rb").read()
    sizes, eof = bgzf_blocks(raw)
    assert eof, "no end-of-file block"
    b = gzip.decompress(raw)
    assert b[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", b, 4)[0]
    out = b[8:8 + l_text].decode()
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", b, at)[0]
    at += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", b, at)[0]
        name = b[at + 4:at + 4 + ln - 1].decode()
        assert b[at + 4 + ln - 1] == 0
        refs.append((name, struct.unpack_from("<i", b, at + 4 + ln)[0]))
        at += 8 + ln
    sq = [l.split("\t") for l in out.splitlines() if l.startswith("@SQ")]
    assert [(dict(f.split(":", 1) for f in l[1:])["SN"], int(dict(f.split(":", 1) for f in l[1:])["LN"])) for l in sq] == refs
    while at < len(b):
        size = struct.unpack_from("<i", b, at)[0]
        rid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, nrid, npos, tlen = struct.unpack_from("<iiBBHHHiiii", b, at + 4)
        p = at + 36
        qname = b[p:p + l_name - 1].decode()
        p += l_name
        cig = ""
        ref_len = 0
        for _ in range(n_cig):
            v = struct.unpack_from("<I", b, p)[0]
            cig += f"{v >> 4}{OPS[v & 15]}"
            if OPS[v & 15] in "MDN=X":
                ref_len += v >> 4
            p += 4
        seq = "".join(SEQ[(b[p + i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = b[p:p + l_seq]
        p += l_seq
        # a read without a position (pos -1) carries reg2bin(-1, 0) = 4680 (SAMv1 section 4.2.1)
        assert bin_ == (4680 if pos < 0 else reg2bin(pos, pos + max(ref_len, 1))), "bin does not match reg2bin"
        tags = []
        end = at + 4 + size
        while p < end:
            tag, typ = b[p:p + 2].decode(), chr(b[p + 2])
            p += 3
            if typ in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
                v = struct.unpack_from(fmt, b, p)[0]
                p += struct.calcsize(fmt)
                tags.append(f"{tag}:i:{v}")
            elif typ == "f":
                tags.append(f"{tag}:f:{struct.unpack_from('<f', b, p)[0]:.6f}")
                p += 4
            elif typ in "ZH":
                e = b.index(0, p)
                tags.append(f"{tag}:{typ}:{b[p:e].decode()}")
                p = e + 1
            elif typ == "A":
                tags.append(f"{tag}:A:{chr(b[p])}")
                p += 1
            else:
                raise AssertionError(f"tag type {typ}")
        assert p == end
        fields = [qname, str(flag), refs[rid][0] if rid >= 0 else "*", str(pos + 1), str(mapq), cig or "*",
                  "*" if nrid < 0 else ("=" if nrid == rid else refs[nrid][0]), str(npos + 1), str(tlen),
                  seq if l_seq else "*", "*" if all(q == 0xFF for q in qual) else "".join(chr(q + 33) for q in qual)]
        out += "\t".join(fields + tags) + "\n"
        at = end
    return out


def reg2bin(beg, end):
    end -= 1
    for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return off + (beg >> sh)
    return 0
