//! T3 parity dumper: witness order / length of the REAL arkworks synthesis against the oracle's golden digests.
//!
//! For every case of `tests/golden/witness_digests.json` (compressed pk / 32-byte message / compressed signature) this
//! builds the circuit exactly as the reference's own test does (src/constraints.rs:335-366: message bytes as witnesses,
//! parameters constant, public key and signature as witnesses, then `verify`), and compares
//!   * cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints(), the result value,
//!   * SHA-256 over the witness assignment, each element as its 6 little-endian u64 MONTGOMERY limbs (48 bytes),
//!   * the same digest per segment of the layout table stored in the golden file,
//! and prints the first segment that differs. With `--params-witness` it does the same for the golden file's "params_witness" section:
//! the circuit with `ParametersVar::new_variable(.., AllocationMode::Witness)` (src/constraints.rs:198-211 takes any mode). With `--pk-input` and / or
//! `--sig-input` it checks the "public_inputs" sections: `PublicKeyVar` / `SignatureVar::new_variable(.., AllocationMode::Input)`
//! (src/constraints.rs:214-249), where cs.instance_assignment (digest `sha256_instance`) carries the point's coordinates. Written for this
//! repository; not derived from the reference's sources beyond calling its public API.
use ark_bls12_381::{Config, Fq};
use ark_crypto_primitives::signature::SigVerifyGadget;
use ark_r1cs_std::alloc::AllocVar;
use ark_r1cs_std::prelude::{AllocationMode, Boolean};
use ark_r1cs_std::uint8::UInt8;
use ark_r1cs_std::R1CSVar;
use ark_relations::r1cs::ConstraintSystem;
use bls_verify_gadget::bls::{Parameters, PublicKey, Signature};
use bls_verify_gadget::constraints::{BlsSignatureVerifyGadget, ParametersVar, PublicKeyVar, SignatureVar};
use sha2::{Digest, Sha256};

fn digest(elems: &[Fq]) -> String {
    let mut h = Sha256::new();
    for x in elems {
        // ark-ff 0.4: Fp(pub BigInt<N>, PhantomData); BigInt(pub [u64; N]) holds the Montgomery representation
        for limb in (x.0).0.iter() {
            h.update(limb.to_le_bytes());
        }
    }
    hex::encode(h.finalize())
}

fn main() {
    let args: Vec<String> = std::env::args().skip(1).collect();
    let params_witness = args.iter().any(|a| a == "--params-witness");
    let pk_input = args.iter().any(|a| a == "--pk-input");
    let sig_input = args.iter().any(|a| a == "--sig-input");
    let path = args.iter().find(|a| !a.starts_with("--")).expect("usage: t3-dumper [--params-witness | --pk-input | --sig-input] <witness_digests.json>");
    let file: serde_json::Value = serde_json::from_str(&std::fs::read_to_string(path).unwrap()).unwrap();
    let golden = if params_witness {
        file["params_witness"].clone()
    } else if pk_input || sig_input {
        let key = format!("pk_{}_sig_{}", if pk_input { "input" } else { "witness" }, if sig_input { "input" } else { "witness" });
        file["public_inputs"][key.as_str()].clone()
    } else {
        file
    };
    let params_mode = if params_witness { AllocationMode::Witness } else { AllocationMode::Constant };
    let pk_mode = if pk_input { AllocationMode::Input } else { AllocationMode::Witness };
    let sig_mode = if sig_input { AllocationMode::Input } else { AllocationMode::Witness };
    let mut all_ok = true;
    for (name, case) in golden["cases"].as_object().unwrap() {
        let cs = ConstraintSystem::<Fq>::new_ref();
        let pk = PublicKey::<Config>::try_from(case["pubkey"].as_str().unwrap()).unwrap();
        let sig = Signature::<Config>::try_from(case["signature"].as_str().unwrap()).unwrap();
        let msg = hex::decode(case["message"].as_str().unwrap()).unwrap();
        let msg_var = UInt8::<Fq>::new_witness_vec(cs.clone(), &msg).unwrap();
        let params = ParametersVar::<Config>::new_variable(cs.clone(), || Ok(Parameters::default()), params_mode).unwrap();
        let pk_var = PublicKeyVar::<Config>::new_variable(cs.clone(), || Ok(pk), pk_mode).unwrap();
        let sig_var = SignatureVar::<Config>::new_variable(cs.clone(), || Ok(sig), sig_mode).unwrap();
        let result: Boolean<Fq> = BlsSignatureVerifyGadget::<Config>::verify(&params, &pk_var, &msg_var, &sig_var).unwrap();

        let inner = cs.borrow().unwrap();
        let w: &Vec<Fq> = &inner.witness_assignment;
        let mut ok = true;
        let mut check = |what: &str, got: String, want: String| {
            if got != want {
                ok = false;
                println!("  MISMATCH {what}: arkworks {got}, oracle {want}");
            }
        };
        check("n_instance_vars", inner.num_instance_variables.to_string(), case["n_instance_vars"].to_string());
        check("n_witness", inner.num_witness_variables.to_string(), case["n_witness"].to_string());
        check("n_constraints", inner.num_constraints.to_string(), case["n_constraints"].to_string());
        check("result", result.value().unwrap().to_string(), case["result"].to_string());
        check("sha256(witness)", digest(w), case["sha256_all"].as_str().unwrap().to_string());
        if let Some(want) = case["sha256_instance"].as_str() {
            check("sha256(instance)", digest(&inner.instance_assignment), want.to_string());
        }
        for seg in golden["segments"].as_array().unwrap() {
            let (seg_name, lo, hi) = (seg[0].as_str().unwrap(), seg[1].as_u64().unwrap() as usize, seg[2].as_u64().unwrap() as usize);
            if hi <= w.len() {
                let want = case["sha256_segments"][seg_name].as_str().unwrap().to_string();
                check(&format!("segment {seg_name} [{lo}, {hi})"), digest(&w[lo..hi]), want);
            }
        }
        println!("{name}: {}", if ok { "IDENTICAL to the oracle (T3 pinned for this case)" } else { "DIFFERS" });
        all_ok &= ok;
    }
    std::process::exit(if all_ok { 0 } else { 1 });
}
