//! `maxvolume` (`src/maxvolume.rs:64-224` of the reference) on top of the HIP-backed `BLU`: one pass over the
//! nonbasic columns of a rectangular matrix, pivoting a column into the basis whenever that multiplies the
//! volume |det B| by more than `volumetol`.  The Python twin `blu_amd/maxvolume.py` is what the tests run.
use crate::{LUInt, Status, BLU};

fn factorize_basis(obj: &mut BLU, a_p: &[usize], a_i: &[usize], a_x: &[f64], basis: &[LUInt]) -> Result<(), Status> {
    // the basis columns as (begin, end) pairs into A (maxvolume.rs:180-197)
    let begin: Vec<usize> = basis.iter().map(|&j| a_p[j as usize]).collect();
    let end: Vec<usize> = basis.iter().map(|&j| a_p[j as usize + 1]).collect();
    obj.factorize(&begin, &end, a_i, a_x)
}

/// Same signature and meaning as `blu::maxvolume`.
pub fn maxvolume(
    obj: &mut BLU,
    ncol: usize,
    a_p: &[usize],
    a_i: &[usize],
    a_x: &[f64],
    basis: &mut [LUInt],
    isbasic: &mut [LUInt],
    volumetol: f64,
    p_nupdate: Option<&mut LUInt>,
) -> Result<(), Status> {
    let mut nupdate: LUInt = 0;
    let result = run(obj, ncol, a_p, a_i, a_x, basis, isbasic, volumetol, &mut nupdate);
    if let Some(p) = p_nupdate {
        *p = nupdate; // written on every exit, as the reference's cleanup label does
    }
    result
}

fn run(
    obj: &mut BLU,
    ncol: usize,
    a_p: &[usize],
    a_i: &[usize],
    a_x: &[f64],
    basis: &mut [LUInt],
    isbasic: &mut [LUInt],
    volumetol: f64,
    nupdate: &mut LUInt,
) -> Result<(), Status> {
    if volumetol < 1.0 {
        return Err(Status::ErrorInvalidArgument);
    }
    let m = basis.len();
    factorize_basis(obj, a_p, a_i, a_x, basis)?;
    for j in 0..ncol {
        if isbasic[j] != 0 {
            continue;
        }
        // lhs = B^-1 a_j
        let (begin, end) = (a_p[j], a_p[j + 1]);
        obj.solve_for_update(end - begin, &a_i[begin..end], Some(&a_x[begin..end]), 'N', 1)?;
        // first entry of largest magnitude, in pattern order
        let (mut xmax, mut xtbl, mut imax) = (0.0f64, 0.0f64, 0usize);
        for k in 0..obj.nzlhs {
            let i = obj.ilhs[k] as usize;
            if obj.lhs[i].abs() > xmax {
                xtbl = obj.lhs[i];
                xmax = xtbl.abs();
                imax = i;
            }
        }
        if xmax <= volumetol {
            continue;
        }
        isbasic[basis[imax] as usize] = 0;
        isbasic[j] = 1;
        basis[imax] = j as LUInt;
        *nupdate += 1;
        obj.solve_for_update(0, &[imax], None, 'T', 0)?;
        obj.update(xtbl)?;
        // refactorize when the eta file is full, the update was inaccurate, or solving has become expensive
        let (nforrest, piverr, cost) = {
            let lu = &obj.lu;
            (lu.nforrest(), lu.pivot_error(), lu.update_cost())
        };
        if nforrest == m || piverr > 1e-8 || cost > 1.0 {
            factorize_basis(obj, a_p, a_i, a_x, basis)?;
        }
    }
    Ok(())
}
