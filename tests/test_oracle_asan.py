"""Sanitizer pass over the oracle's C restatement (CPU build only: GPU AddressSanitizer is not
available on this pool): AddressSanitizer + UBSan on PDHG, gap, and the banded adjoint solve,
including ragged shapes."""
import os
import subprocess
import pytest
from conftest import ROOT

DRIVER = r'''
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
int bplo_pdhg(int,int,int,const double*,const double*,int,int,double,double,double,int,int,double*,double*,double*,int);
double bplo_cost(int,int,int,const double*,const double*,double*);
void bplo_gap(int,int,int,const double*,const double*,const double*,const double*,const double*,int,int,double*);
int bplo_gradient(int,int,int,const double*,const double*,const double*,int,int,int,double,int,double*,double*);
int main(void){
  const int shapes[4][3]={{13,7,2},{1,9,1},{8,1,1},{24,20,3}};
  for(int s=0;s<4;++s){ int M=shapes[s][0],N=shapes[s][1],O=shapes[s][2]; size_t n=(size_t)M*N*O;
    double *f=malloc(n*8),*ub=malloc(n*8),*x=malloc(n*8),*y1=malloc(n*8),*y2=malloc(n*8),*gap=malloc(O*8);
    unsigned r=12345u+s; for(size_t k=0;k<n;++k){ r=r*1664525u+1013904223u; f[k]=(r>>8)/16777216.0; ub[k]=0.5*f[k]+0.25; }
    double a1=0.1, a22[4]={0.05,0.1,0.2,0.08}, g[4];
    if(bplo_pdhg(M,N,O,f,&a1,1,1,0.0,5.0,0.198,1,60,x,y1,y2,1)) return 1;
    bplo_gap(M,N,O,x,y1,y2,f,&a1,1,1,gap);
    if(bplo_gradient(M,N,O,x,ub,&a1,1,1,0,1e14,3,g,NULL)) return 2;
    if(bplo_gradient(M,N,O,x,ub,&a1,1,1,1,1e14,3,g,NULL)) return 3;
    if(M>=2&&N>=2){ if(bplo_pdhg(M,N,O,f,a22,2,2,0.01,5.0,0.198,0,40,x,NULL,NULL,1)) return 4;
      if(bplo_gradient(M,N,O,x,ub,a22,2,2,0,1e14,2,g,NULL)) return 5;
      if(bplo_gradient(M,N,O,x,ub,a22,2,2,1,1e14,2,g,NULL)) return 6; }
    printf("shape %dx%dx%d cost %.6f gap %.3e grad %.6f\n",M,N,O,bplo_cost(M,N,O,x,ub,NULL),gap[0],g[0]);
    free(f);free(ub);free(x);free(y1);free(y2);free(gap); }
  return 0; }
'''


def test_oracle_under_asan_ubsan(tmp_path):
    src = tmp_path / "drv.c"
    src.write_text(DRIVER)
    exe = tmp_path / "drv"
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off",
           "-mfma", "-o", str(exe), str(src), os.path.join(ROOT, "oracle", "bpltv_oracle.c"), "-lm"]
    try:
        subprocess.check_call(cmd)
    except (subprocess.CalledProcessError, FileNotFoundError) as e:
        pytest.skip("sanitizer build unavailable: %s" % e)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr
    assert out.stdout.count("shape") == 4
