/* Developer probe (CPU, no GPU needed): how good a GUESS is the cheap warm-up of the raw-composite decoder's second
 * sweep (csrc/raw28_decode.hip, k_raw28_follow), and how many scanlines of exact steps does it take after it until the
 * level is bit-identical to the serial walk?
 *   gcc -O2 -ffp-contract=off tools/follow_guess_probe.c -lm -o /tmp/fgp
 *   python -c "import sys; sys.path[:0]=['tests']; import _libs as L; L.raw28_capture(24, 5, 3, 0).tofile('/tmp/cap.bin')"
 *   /tmp/fgp /tmp/cap.bin [samples per superblock = 64] [margin = 0.0625]
 * Arithmetic of hsync_dc_proc() as in oracle/raw28_oracle.c (reference: ffmpeg_raw28ntsc.cpp:556-594); the closed form
 * of a superblock is level = fma(level, om_slow^SB, a_slow * SUM lv_i om_slow^(SB-1-i)) where level + margin < min lv. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET);
  uint8_t*raw=malloc(n); fread(raw,1,n,f); fclose(f);
  int SB=argc>2?atoi(argv[2]):64; double margin=argc>3?atof(argv[3]):0.0625;
  double rate=(315000000.00*8.0)/88.00, frame_t=rate/(30000.00/1001.00), line_t=frame_t/525.00;
  double hz=rate/(line_t*0.075*0.75), tau=1/(hz*2*M_PI), ti=1.0/rate, alpha=ti/(tau+ti);
  double aF=1.0/(line_t*0.07*0.75), omF=1.0-aF, aS=1.0/(frame_t*0.6), omS=1.0-aS;
  double prev=0; for(size_t j=0;j<frame_t;j++){double s1=128*alpha,s2=prev-(prev*alpha);prev=s1+s2;}
  double p0=prev,p1=prev,p2=prev;
  double*lv=malloc(sizeof(double)*n), *tr=malloc(sizeof(double)*n);
  double level=128.0;
  for(long i=0;i<n;i++){double x=raw[i];
    x=(x*alpha)+(p0-(p0*alpha));p0=x; x=(x*alpha)+(p1-(p1*alpha));p1=x; x=(x*alpha)+(p2-(p2*alpha));p2=x; lv[i]=x;
    if(level>x) level=(level*(1.0-aF))+(x*aF); else level=(level*(1.0-aS))+(x*aS);
    tr[i]=level;}
  long nsb=n/SB; double*bmin=malloc(8*nsb),*B=malloc(8*nsb);
  for(long s=0;s<nsb;s++){double T=0,mn=1e300;for(int j=0;j<SB;j++){double x=lv[s*SB+j];T=fma(T,omS,x);if(x<mn)mn=x;}bmin[s]=mn;B[s]=T*aS;}
  double omSB=1; for(int j=0;j<SB;j++)omSB*=omS;
  int LEN=1820; int WL=112;
  // for several chunk starts g: cheap from g-WL*LEN (level 255) for (WL-E) lines then exact; report merge line
  long tot_ok=0,tot_non=0; int worst_merge=0; double worst_err=0;
  int nst=0; long hist[64]; memset(hist,0,sizeof hist);
  for(long g=(long)WL*LEN+64*1000; g+40*LEN<n; g+= 17472){
    long s0=(g-(long)WL*LEN)/SB; // superblock aligned start
    for(int CL=60; CL<=100; CL+=40){ // cheap lines
    double L=255.0; long s=s0; long cheap_end=(s0*SB+(long)CL*LEN)/SB;
    for(;s<cheap_end;s++){
      if(L+margin<bmin[s]){L=fma(L,omSB,B[s]);tot_ok++;}
      else{tot_non++;for(int j=0;j<SB;j++){double x=lv[s*SB+j]; double dl=x-L; L=fma(dl,(dl<0?aF:aS),L);}}   /* the kernel's guess_block: one fma per step */
    }
    long i=s*SB; double err=fabs(L-tr[i-1]);
    if(CL==100 && err>worst_err)worst_err=err;
    long i0=i; int merged=-1;
    for(;i<n && i<i0+60L*LEN;i++){double x=lv[i]; if(L>x)L=(L*(1.0-aF))+(x*aF); else L=(L*(1.0-aS))+(x*aS);
      if(memcmp(&L,&tr[i],8)==0){ // require it stays merged (it will, deterministic)
        merged=(int)((i-i0)/LEN); break;}}
    if(CL==100){ if(merged<0)merged=63; hist[merged]++; if(merged>worst_merge)worst_merge=merged; nst++;}
    else if(g<(long)WL*LEN+64*1000+17472*3) printf("CL=60 g=%ld err_at_switch=%.3e merged_after_lines=%d\n",g,err,merged);
    }
  }
  printf("SB=%d margin=%g: starts=%d ok_blocks=%ld nonok=%ld (%.1f%% nonok) worst err after 100 cheap lines=%.3e worst merge=%d lines\n",SB,margin,nst,tot_ok,tot_non,100.0*tot_non/(tot_ok+tot_non),worst_err,worst_merge);
  for(int k=0;k<64;k++) if(hist[k]) printf("  merge after %d lines: %ld\n",k,hist[k]);
  return 0;}
