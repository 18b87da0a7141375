"""Writers of the on-disk formats the library reads -- the counterpart of the reference's offline converters.

  convert_ev_to_binary(file, read_as)          script/convert_ev_to_binary.py:72-140 (the __main__ body)
      read_as "fp32"     write_as_binary        :58-69   raw native-endian fp32, every column but the last ("key")
      read_as "u_short"  write_ushort_per_byte  :43-56   struct 'H' (native endian) per element
      read_as "u_char"   write_uchar_per_byte   :31-41   struct '>B' per element (u8 codes, or packed u4 pairs)
  convert_altkeys_to_binary(input, output)     script/convert_altkeys_to_binary.py:39-57
      one "T-R" line per row  ->  4-byte BIG-endian  R * 100 + T   (:27-37, '>I')
  convert_altkeys_folder(input_folder)         script/convert_altkeys_to_binary.py:59-84 (the __main__ body)
  write_altkeys(table_ids, row_ids, path)      the same file from arrays (what a GPU k-NN pass would hand over)
  tables_to_bin_dir(ev, out_dir)               EVTables (any precision, HBM or pinned) -> ev-table-{1..T}.bin

Output PLACEMENT follows the reference: fp32 goes to <dir of the csv>/binary/<name>.bin, u_short / u_char to
<PARENT of that dir>/binary/<name>.bin (:97 vs :110-117, :135-140).  The reference walks the DataFrame element by
element through struct.pack; here a table is one numpy conversion and one write -- the bytes are the same
(tests/test_converters.py against files written by the reference's own functions, tests/golden/converters.npz).
"""
import os
import sys
from pathlib import Path

import numpy as np

OUT_BINARY_DIR_NAME = "binary/"


def _read_csv(path, delimiter=","):
    """The reference reads with pandas (header row = column names, dtype=object then astype).  The files are plain
    numeric CSVs with one header line (dlrm_s_pytorch.py:1787-1792: "0,1,...,d-1"), so numpy's reader sees the same
    cells; the header is returned because the fp32 writer keys off a column called 'key'."""
    with open(path) as f:
        header = f.readline().rstrip("\r\n").split(delimiter)
        rows = [line.rstrip("\r\n").split(delimiter) for line in f if line.strip()]
    return header, rows


def _fp32_bytes(header, rows):
    # convert_ev_to_binary.py:99-104: astype(np.float32); a missing 'key' column is appended (= the index); then
    # write_as_binary (:58-69) emits every column EXCEPT THE LAST, each value's 4 raw bytes
    a = np.array(rows, dtype=object)
    vals = np.empty(a.shape, np.float32)
    if a.size:
        # string -> float64 -> float32: what DataFrame.astype(np.float32) does on object cells
        vals = np.asarray(a, dtype=np.float64).astype(np.float32)
    if "key" in header:
        vals = vals[:, :-1]  # the reference drops the LAST column, wherever 'key' sits
    return np.ascontiguousarray(vals).tobytes()


def _int_cells(rows):
    # .astype(np.int) on object cells holding integer literals (the reduced-precision CSVs of reduce_precision.py)
    return np.array([[int(c) for c in r] for r in rows], dtype=np.int64).reshape(len(rows), -1)


def convert_ev_to_binary(file, read_as, verbose=True):
    """-> path of the .bin written.  Errors print and exit(-1) like the reference."""
    parent = str(Path(file).parent)  # pathlib like the reference: Path("t.csv").parent == ".", Path(".").parent == "."
    name = os.path.splitext(os.path.basename(file))[0]
    if verbose:
        print("===== Read ev data as " + str(read_as))
    if read_as == "fp32":
        out_dir = os.path.join(parent, OUT_BINARY_DIR_NAME)
        header, rows = _read_csv(file)
        blob = _fp32_bytes(header, rows)
    elif read_as in ("u_short", "u_char"):
        out_dir = os.path.join(str(Path(parent).parent), OUT_BINARY_DIR_NAME)
        _, rows = _read_csv(file)
        cells = _int_cells(rows)
        lim = 65535 if read_as == "u_short" else 255
        if cells.size and (cells.min() < 0 or cells.max() > lim):
            # struct.pack('H' / '>B', v) raises struct.error on such a value; fail as loudly
            print("ERROR: value out of range for " + read_as + " in " + file)
            sys.exit(-1)
        blob = cells.astype(np.uint16 if read_as == "u_short" else np.uint8).tobytes()  # 'H' is NATIVE endian (:52)
    elif read_as == "fp16":
        print("ERROR: read as fp16 is no longer supported! It's too slow to unpack by C++")
        sys.exit(-1)
    else:
        print("ERROR: Can't understand the read_as format : " + str(read_as))
        sys.exit(-1)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".bin")
    with open(out, "wb") as f:
        f.write(blob)
    if verbose:
        print("===== output file : " + out)
        print("Done")
    return out


def altkey_words(table_ids, row_ids):
    """alt key = row * 100 + table (table 1-based, two decimal digits: convert_altkeys_to_binary.py:43-49), as the
    big-endian u32 array the tier's loader reads (aprx_embedding.cpp:243-251)."""
    t = np.asarray(table_ids, dtype=np.int64)
    r = np.asarray(row_ids, dtype=np.int64)
    k = t + 100 * r
    if k.size and (k.min() < 0 or k.max() > 0xFFFFFFFF):
        print("ERROR: alt key does not fit 4 bytes")  # struct.pack('>I') raises
        sys.exit(-1)
    return k.astype(">u4")


def write_altkeys(table_ids, row_ids, path):
    with open(path, "wb") as f:
        f.write(altkey_words(table_ids, row_ids).tobytes())
    return path


def convert_altkeys_to_binary(input_file, output_file, verbose=True):
    """One 'tableId-rowId' line per row of the table (no header: read_csv(..., delimiter='-', header=None), :40)."""
    tids, rids = [], []
    with open(input_file) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            a, b = line.split("-")[:2]
            tids.append(int(a))
            rids.append(int(b))
    write_altkeys(tids, rids, output_file)
    if verbose:
        print("===== output file : " + output_file)
    return output_file


def convert_altkeys_folder(input_folder, verbose=True):
    """Every file of the folder whose name contains 'ev-table' -> <folder>/binary/<name>.bin (:70-84)."""
    files = [os.path.join(input_folder, n) for n in os.listdir(input_folder)
             if os.path.isfile(os.path.join(input_folder, n)) and "ev-table" in n]
    if not files:
        print("ERROR: Can't find files (*ev-table*) in folder: " + input_folder)
        sys.exit(-1)
    os.makedirs(os.path.join(input_folder, OUT_BINARY_DIR_NAME), exist_ok=True)
    outs = []
    for p in files:
        if verbose:
            print("Processing ... " + p)
        name = os.path.splitext(os.path.basename(p))[0]
        outs.append(convert_altkeys_to_binary(p, os.path.join(input_folder, OUT_BINARY_DIR_NAME, name + ".bin"), verbose))
    return outs


def tables_to_bin_dir(ev, out_dir):
    """EVTables -> out_dir/ev-table-{1..T}.bin.  The tables already sit in HBM in the on-disk byte layout (docs/HISTORY.md 2),
    so this is one device-to-host copy and one write per table, whatever the precision; chained behind
    EVTables.encode(bits) it is reduce_precision.py + convert_ev_to_binary.py without the CSV round trip."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for k, t in enumerate(ev.raw):
        p = os.path.join(out_dir, "ev-table-%d.bin" % (k + 1))
        a = t.detach().cpu().contiguous().numpy()
        with open(p, "wb") as f:
            f.write(memoryview(a.reshape(-1)))
        paths.append(p)
    return paths


def _main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="CSV -> .bin (convert_ev_to_binary.py) / alt-key folder -> .bin")
    ap.add_argument("-file", type=str, help="File path of the raw ev data")
    ap.add_argument("-read_as", type=str, help="fp32 | u_short | u_char")
    ap.add_argument("-input_folder", type=str, help="Folder path of the raw alternative-keys files")
    a = ap.parse_args(argv)
    if a.input_folder:
        convert_altkeys_folder(a.input_folder)
        print("output folder : " + a.input_folder + "/binary/")
        print("Done")
    elif a.file and a.read_as:
        convert_ev_to_binary(a.file, a.read_as)
    else:
        print("ERROR: You must provide these 2 arguments: -file <the input file> -read_as <read as fp32 or fp16>  ")
        sys.exit(-1)


if __name__ == "__main__":
    _main()
