//! Dumps golden vectors from the real crates (simd-minimizers 1.3.0, packed-seq 3.2.1, xxhash-rust 0.8.15).
//!
//! The body of `hashes_and_positions` makes the same calls, in the same order, as the reference's
//! `get_minimizer_hashes_and_positions` (src/filter_common.rs:211-310); only the effective-sequence
//! handling (prefix cut, newline strip) is left out, because the vectors carry plain sequences.
//! `raw_positions` is the crate's answer before the reference's ACGT filter, which is what separates
//! the candidate rules (rotation 1|7, 16|32 compared bits, + | ^) most directly.
//!
//! Two more sections come from the reference's OWN public functions (the `deacon` library crate, src/lib.rs:26-33, 276-286),
//! so nothing of them is restated here:
//!   * `index_vectors`: `deacon::compute_minimizer_hashes(seq, k, w, entropy_threshold)` = fill_minimizer_hashes,
//!     src/minimizers.rs:125-191 (IUPAC rewrite, `AsciiSeq` view, ACGT test on the original bytes, entropy floor :73-121);
//!   * `index_file`: the bytes `deacon::write_minimizers` (src/index.rs:130-164: bincode 2.0.1 `encode_into_std_write`,
//!     `config::standard()`) writes for a small key set that holds every varint width and its boundary values.
//!
//! Output: one JSON document on stdout, schema of tests/golden/oracle_vectors.json plus `raw_positions`:
//!   {"source": "...", "vectors": [{"k":31,"w":15,"seq":"ACGT...","raw_positions":[..],"positions":[..],"hashes":["0x.."]}],
//!    "index_vectors": [{"k":31,"w":15,"entropy_threshold":"0.5","seq":"ACGTN...","hashes":["0x.."]}],
//!    "index_file": {"k":31,"w":15,"keys_sorted":["0x0",..],"hex":"021f0f.."}}
//!
//! Sequences: every `seq` of ../oracle_vectors.json (read at run time, so the two files cannot drift), the
//! reference's own test literals (tests/filter_tests.rs:43-63, 957-966, 1203), and a seeded set of longer reads.

use packed_seq::SeqVec;
use rustc_hash::FxHashSet;
use std::fmt::Write as _;

fn hashes_and_positions(seq: &[u8], k: usize, w: usize) -> (Vec<u32>, Vec<u32>, Vec<u64>) {
    // src/filter_common.rs:238
    let packed_seq = packed_seq::PackedSeqVec::from_ascii(seq);
    // src/filter_common.rs:245-258
    let mut invalid_mask = vec![0u64; packed_seq.len() / 64 + 2];
    for i in (0..seq.len()).step_by(64) {
        let mut mask = 0u64;
        for (j, b) in seq[i..(i + 64).min(seq.len())].iter().enumerate() {
            mask |= ((!matches!(b, b'A' | b'C' | b'G' | b'T' | b'a' | b'c' | b'g' | b't')) as u64) << j;
        }
        invalid_mask[i / 64] = mask;
    }
    // src/filter_common.rs:261-267
    let mut positions: Vec<u32> = Vec::new();
    simd_minimizers::canonical_minimizer_positions(packed_seq.as_slice(), k, w, &mut positions);
    let raw = positions.clone();
    // src/filter_common.rs:275-286
    assert!(k <= 56);
    positions.retain(|&pos| {
        let mask = u64::MAX >> (64 - k);
        let byte = pos as usize / 8;
        let offset = pos as usize % 8;
        let x = (unsafe { invalid_mask.as_ptr().byte_add(byte).read_unaligned() } >> offset) & mask;
        x == 0
    });
    // src/filter_common.rs:289-307
    let mut hashes: Vec<u64> = Vec::new();
    if k > 32 {
        hashes.extend(
            simd_minimizers::iter_canonical_minimizer_values_u128(packed_seq.as_slice(), k, &positions)
                .map(|kmer| xxhash_rust::xxh3::xxh3_64(&kmer.to_le_bytes())),
        );
    } else {
        hashes.extend(
            simd_minimizers::iter_canonical_minimizer_values(packed_seq.as_slice(), k, &positions)
                .map(|kmer| xxhash_rust::xxh3::xxh3_64(&kmer.to_le_bytes())),
        );
    }
    (raw, positions, hashes)
}

/// splitmix64: the seeded reads below must be reproducible without any crate
fn splitmix(state: &mut u64) -> u64 {
    *state = state.wrapping_add(0x9E3779B97F4A7C15);
    let mut z = *state;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^ (z >> 31)
}

fn random_seq(state: &mut u64, n: usize, p_n_per_1024: u64, lower_per_1024: u64) -> String {
    let mut s = String::with_capacity(n);
    for _ in 0..n {
        let r = splitmix(state);
        let mut c = [b'A', b'C', b'G', b'T'][(r & 3) as usize];
        if (r >> 8) % 1024 < p_n_per_1024 {
            c = b'N';
        } else if (r >> 24) % 1024 < lower_per_1024 {
            c |= 0x20;
        }
        s.push(c as char);
    }
    s
}

/// every "seq": "..." / "k": n / "w": n triple of ../oracle_vectors.json, without a JSON crate: the file is
/// written by tests/golden/make_golden.py with json.dump(indent=1), one key per line
fn vectors_of_oracle_file(path: &str) -> Vec<(usize, usize, String)> {
    let text = match std::fs::read_to_string(path) {
        Ok(t) => t,
        Err(_) => return Vec::new(),
    };
    let (mut k, mut w, mut out) = (0usize, 0usize, Vec::new());
    for line in text.lines() {
        let t = line.trim().trim_end_matches(',');
        if let Some(v) = t.strip_prefix("\"k\": ") {
            k = v.parse().unwrap_or(0);
        } else if let Some(v) = t.strip_prefix("\"w\": ") {
            w = v.parse().unwrap_or(0);
        } else if let Some(v) = t.strip_prefix("\"seq\": ") {
            out.push((k, w, v.trim_matches('"').to_string()));
        }
    }
    out
}

/// IUPAC codes, both cases, and a few bytes outside the alphabet: what src/minimizers.rs:24-43 rewrites
fn random_iupac(state: &mut u64, n: usize, ambiguous_per_1024: u64, lower_per_1024: u64) -> String {
    const AMBIG: &[u8] = b"RYSWKMBDHVNX-";
    let mut s = String::with_capacity(n);
    for _ in 0..n {
        let r = splitmix(state);
        let mut c = [b'A', b'C', b'G', b'T'][(r & 3) as usize];
        if (r >> 8) % 1024 < ambiguous_per_1024 {
            c = AMBIG[((r >> 40) % AMBIG.len() as u64) as usize];
        }
        if c.is_ascii_alphabetic() && (r >> 24) % 1024 < lower_per_1024 {
            c |= 0x20;
        }
        s.push(c as char);
    }
    s
}

/// low-complexity stretches between random ones: minimizers on both sides of an entropy floor
fn low_complexity(state: &mut u64, n: usize) -> String {
    let mut s = String::with_capacity(n);
    while s.len() < n {
        let r = splitmix(state);
        let run = 20 + (r >> 8) as usize % 60;
        match r & 3 {
            0 => s.push_str(&"A".repeat(run)),
            1 => s.push_str(&"AC".repeat(run / 2)),
            2 => s.push_str(&"AAAT".repeat(run / 4)),
            _ => s.push_str(&random_seq(state, run, 0, 0)),
        }
    }
    s.truncate(n);
    s
}

/// index side: the reference's own fill_minimizer_hashes (src/minimizers.rs:125-191) through its public re-export
fn index_vectors(out: &mut String) {
    let mut st = 20261005u64;
    let mut cases: Vec<(u8, u8, &str, String)> = Vec::new(); // k, w, entropy threshold as text, sequence
    for &(k, w) in &[(31u8, 15u8), (15, 11), (41, 15), (21, 9), (9, 5), (32, 16), (56, 2)] {
        for &thr in &["0.0", "0.5"] {
            cases.push((k, w, thr, random_iupac(&mut st, 400, 0, 0)));
            cases.push((k, w, thr, random_iupac(&mut st, 1500, 24, 200)));
            cases.push((k, w, thr, low_complexity(&mut st, 1200)));
        }
    }
    cases.push((31, 15, "0.25", low_complexity(&mut st, 3000)));
    cases.push((31, 15, "0.9", random_iupac(&mut st, 3000, 8, 0)));
    cases.push((31, 15, "0.0", "ACGTNNNNACGT".repeat(12)));           // non-ACGT k-mers on the original bytes
    cases.push((31, 15, "0.0", random_iupac(&mut st, 40, 0, 0)));     // k <= len < k+w-1: no window
    cases.push((31, 15, "0.0", random_iupac(&mut st, 20, 0, 0)));     // len < k (:136-139)
    cases.push((31, 15, "0.0", random_iupac(&mut st, 70_000, 2, 20))); // beyond 65,536 bases
    out.push_str(" \"index_vectors\": [\n");
    for (i, (k, w, thr, seq)) in cases.iter().enumerate() {
        let t: f32 = thr.parse().unwrap();
        let hashes = deacon::compute_minimizer_hashes(seq.as_bytes(), *k, *w, t);
        let hx = hashes.iter().map(|h| format!("\"{:#x}\"", h)).collect::<Vec<_>>().join(", ");
        let _ = write!(
            out,
            "  {{\"k\": {}, \"w\": {}, \"entropy_threshold\": \"{}\", \"seq\": \"{}\", \"hashes\": [{}]}}{}\n",
            k, w, thr, seq, hx, if i + 1 == cases.len() { "" } else { "," }
        );
    }
    out.push_str(" ],\n");
}

/// index file: the reference's own writer (src/index.rs:130-164 through src/lib.rs:280-286), bytes as hex
fn index_file(out: &mut String) {
    let mut keys: Vec<u64> = vec![0, 1, 250, 251, 252, 65_535, 65_536, 0xFFFF_FFFF, 0x1_0000_0000, u64::MAX, u64::MAX - 1,
                                  0xC77B3ABB6F87ACD9, 0x2FBC593564DB792E]; // (the last two: XXH3 of k-mer values 0 and 1)
    let mut st = 20261006u64;
    for _ in 0..300 {
        keys.push(splitmix(&mut st));               // 9-byte encodings, as nearly all real hashes
    }
    for _ in 0..20 {
        keys.push(splitmix(&mut st) >> 40);          // 5-byte encodings
        keys.push(splitmix(&mut st) >> 52);          // 3-byte and 1-byte encodings
    }
    let set: FxHashSet<u64> = keys.iter().copied().collect();
    let header = deacon::IndexHeader::new(31, 15);
    let path = std::env::temp_dir().join(format!("dump_crate_vectors_{}.idx", std::process::id()));
    deacon::write_minimizers(&set, &header, Some(&path)).expect("write_minimizers");
    let bytes = std::fs::read(&path).expect("read the index file back");
    // and what the reference's own loader makes of it (src/index.rs:80-107)
    let (loaded, h2) = deacon::load_minimizers(&path).expect("load_minimizers");
    let _ = std::fs::remove_file(&path);
    let loaded = loaded.expect("a key set");
    assert!(loaded == set && h2.kmer_length() == 31 && h2.window_size() == 15);
    let mut sorted: Vec<u64> = set.iter().copied().collect();
    sorted.sort_unstable();
    let kx = sorted.iter().map(|h| format!("\"{:#x}\"", h)).collect::<Vec<_>>().join(", ");
    let hex = bytes.iter().map(|b| format!("{:02x}", b)).collect::<String>();
    let _ = write!(out, " \"index_file\": {{\"k\": 31, \"w\": 15, \"keys_sorted\": [{}], \"hex\": \"{}\"}}\n", kx, hex);
}

fn main() {
    let mut cases: Vec<(usize, usize, String)> = vectors_of_oracle_file("../oracle_vectors.json");
    // the reference's own literals
    let sc2_0_60 = "ATTAAAGGTTTATACCTTCCCAGGTAACAAACCAACCAACTTTCGATCTCTTGTAGATCT"; // tests/filter_tests.rs:43-47
    let sc2_0_60_rev = "AGATCTACAAGAGATCGAAAGTTGGTTGGTTTGTTACCTGGGAAGGTATAAACCTTTAAT"; // :57-63
    let pair1 = format!("{}{}{}", "A".repeat(9), "ACGT".repeat(16), "A".repeat(10)); // :957-966
    let pair2 = format!("{}{}{}", "T".repeat(10), "ACGT".repeat(16), "T".repeat(10));
    cases.push((31, 15, sc2_0_60.to_string()));
    cases.push((31, 15, sc2_0_60_rev.to_string()));
    cases.push((31, 15, "ACGT".repeat(28)));
    cases.push((31, 15, pair1));
    cases.push((31, 15, pair2));
    cases.push((5, 5, "AAAAACAAAAACAAAAACAAAAA".to_string())); // :1203
    cases.push((5, 5, "A".repeat(20)));
    // seeded reads: long enough that a wrong rotation / compare width / combination cannot hide, several (k, w),
    // with N and lower case, and one read beyond 65,536 bases (the crate's 16-bit position packing)
    let mut st = 20261004u64;
    for &(k, w) in &[(31usize, 15usize), (31, 15), (15, 11), (41, 15), (21, 9), (31, 1), (13, 7), (27, 19), (56, 2), (32, 16), (7, 3)] {
        for &(n, pn, pl) in &[(150usize, 0u64, 0u64), (1000, 0, 0), (3000, 8, 100)] {
            cases.push((k, w, random_seq(&mut st, n, pn, pl)));
        }
    }
    cases.push((31, 15, random_seq(&mut st, 70_000, 1, 0)));

    let mut out = String::new();
    out.push_str("{\n \"source\": \"simd-minimizers 1.3.0 + packed-seq 3.2.1 + xxhash-rust 0.8.15, calls of src/filter_common.rs:238-307; index_vectors / index_file: the deacon crate's own compute_minimizer_hashes / write_minimizers (bincode 2.0.1)\",\n \"vectors\": [\n");
    for (i, (k, w, seq)) in cases.iter().enumerate() {
        if (k + w - 1) % 2 == 0 || seq.len() < k + w - 1 {
            continue;
        }
        let (raw, pos, hashes) = hashes_and_positions(seq.as_bytes(), *k, *w);
        let join = |v: &Vec<u32>| v.iter().map(|x| x.to_string()).collect::<Vec<_>>().join(", ");
        let hx = hashes.iter().map(|h| format!("\"{:#x}\"", h)).collect::<Vec<_>>().join(", ");
        let _ = write!(
            out,
            "  {{\"k\": {}, \"w\": {}, \"seq\": \"{}\", \"raw_positions\": [{}], \"positions\": [{}], \"hashes\": [{}]}}{}\n",
            k, w, seq, join(&raw), join(&pos), hx, if i + 1 == cases.len() { "" } else { "," }
        );
    }
    out.push_str(" ],\n");
    // a trailing comma before the closing bracket (when the last case was skipped) would not be JSON
    let mut out = out.replace(",\n ],", "\n ],");
    index_vectors(&mut out);
    index_file(&mut out);
    out.push_str("}\n");
    print!("{}", out);
}
