"""CPU oracle for the Marlin prover hot path (MSM + NTT inside Marlin::prove).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import or execute it, and there only as the checker.  The product path
(``marlin_amd``) never imports this package and fails loudly when the HIP
library is missing.

Parity status: **parity unpinned against arkworks bytes**.  The reference
(/root/reference, arkworks-rs/marlin 0.3.0) holds no golden vectors or
known-answer tests for this path, its arithmetic lives in
crates.io dependencies that are absent from the container (ark-ff / ark-ec /
ark-poly / ark-poly-commit ^0.3.0, Cargo.toml:23-28, no Cargo.lock), and no Rust
toolchain exists here.  The oracle is therefore pinned by
 (1) mathematical uniqueness (NTT == naive DFT, MSM == naive sum of scalar muls),
 (2) public curve/field known answers (generator on curve, [r]G = O, root of
     unity orders, Montgomery constants: SURVEY.md Appendix D),
 (3) known-tau KZG identities (commit(p) == [p(tau)]G), and
 (4) the reference's own test *properties* (prove->verify accepts / rejects,
     sumcheck LCs evaluate to zero, degree asserts: src/test.rs:158-161,
     src/ahp/mod.rs:177,214, src/ahp/prover.rs:385-388,516,556-557,697-698).
"""
