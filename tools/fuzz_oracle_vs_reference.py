#!/usr/bin/env python3
"""Fuzz the C oracle against the reference itself (build container only: needs /root/reference): random frames of random
lengths, signal kinds and sample rates through NFM / AM / WFM / iq_correction / power, bit for bit.
    python tools/fuzz_oracle_vs_reference.py        # prints the case counts and the number of mismatches
A run of 240 cases per function found the np.var form used by iq_correction (squares + add, no FMA); clean since."""
import sys, warnings; sys.path.insert(0,'/root/reference'); import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')); sys.dont_write_bytecode=True
import numpy as np, signal_processing as sp, scipy.signal as ss
import oracle_lib as O
warnings.simplefilter('ignore')
rng=np.random.default_rng(int(os.environ.get("FUZZ_SEED", "2026")))
def rnd_iq(n):
    kind=rng.integers(0,4)
    if kind==0:
        ph=np.cumsum(rng.standard_normal(n)*rng.uniform(0.01,0.5)); x=rng.uniform(0.05,2.0)*np.exp(1j*ph)
    elif kind==1:
        x=rng.standard_normal(n)+1j*rng.standard_normal(n)
    elif kind==2:
        u=rng.integers(0,256,size=(n,2)); x=((u[:,0]-127.5)/127.5)+1j*((u[:,1]-127.5)/127.5)
    else:
        t=np.arange(n); x=0.3*np.exp(2j*np.pi*rng.uniform(-0.4,0.4)*t)+0.01*(rng.standard_normal(n)+1j*rng.standard_normal(n))
    x=x+rng.uniform(0,0.05)*(rng.standard_normal(n)+1j*rng.standard_normal(n))
    return x.astype(np.complex64)
bad=0; cnt={'nfm':0,'am':0,'wfm':0,'iqc':0,'pow':0}
rates=[2.4e6,1.024e6,2.048e6,250e3,3.2e6,10e6]
for it in range(int(os.environ.get("FUZZ_N", "240"))):
    n=int(rng.choice(eval(os.environ.get("FUZZ_SIZES", "[29,30,64,100,257,1000,1024,2048,4097,8192,16384,20000,33000]"))))
    fs=float(rng.choice(rates)); x=rnd_iq(n); q=int(fs/22050)
    # NFM
    taps=ss.firwin(65,15000/(fs/2)); sos=ss.cheby1(8,0.05,0.8/q,output='sos'); zi=ss.sosfilt_zi(sos)
    ref=sp.demodulate_signal(x,fs,'NFM')[:,0]; got=O.demod_nfm(x,fs,taps,sos,zi)
    ok=np.array_equal(ref,got,equal_nan=True); cnt['nfm']+=1
    if not ok: bad+=1; print('NFM mismatch',n,fs,np.nanmax(np.abs(ref-got)))
    # AM
    am=ss.butter(5,[300/11025,3000/11025],btype='band',output='sos')
    ref=sp.demodulate_signal(x,fs,'AM')[:,0]; got=O.demod_am(x,am); cnt['am']+=1
    if not np.array_equal(ref,got,equal_nan=True): bad+=1; print('AM mismatch',n,fs)
    # iq_correction
    ref=sp.iq_correction(x); got=O.iq_correction(x); cnt['iqc']+=1
    if not np.array_equal(ref.view(np.uint32),got.view(np.uint32)): bad+=1; print('IQC mismatch',n, np.mean(ref.view(np.uint32)!=got.view(np.uint32)))
    # power
    ref=sp.measure_signal_power(x); got=O.power_db(x); cnt['pow']+=1
    if np.float32(ref).tobytes()!=np.float32(got).tobytes() and not (np.isnan(ref) and np.isnan(got)): bad+=1; print('POW mismatch',n,ref,got)
    # WFM
    if fs>106e3+1 and q>=2:
        nyq=fs/2
        filt=dict(lp_sos=ss.butter(5,15000/nyq,btype='low',output='sos'),pilot_sos=ss.butter(5,[18800/nyq,19200/nyq],btype='band',output='sos'),lmr_sos=ss.butter(5,[23000/nyq,53000/nyq],btype='band',output='sos'),alpha=np.exp(-1/(75e-6*fs)),dec_sos=sos,dec_zi=zi)
        ref=sp.demodulate_signal(x,fs,'WFM'); got=O.demod_wfm(O.iq_correction(x),fs,filt); cnt['wfm']+=1
        if not np.array_equal(ref.view(np.uint64),got.view(np.uint64)):
            nanok=np.array_equal(ref,got,equal_nan=True)
            if not nanok: bad+=1; print('WFM mismatch',n,fs,np.nanmax(np.abs(ref-got)))
print('cases',cnt,'bad',bad)
