R=$GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$R/opensmile_amd:$R/oracle/_ref
cd $R/opensmile_amd/plugin
SMILEHIP_PLUGIN_FUSE=1 $R/oracle/_ref/SMILExtract -C $R/oracle/_ref/config/mfcc/MFCC12_0_D_A.conf -I $R/tests/golden/files/u3_4000.wav -O /tmp/o.htk -l 2 2>&1 | grep -v "plugin '" | tail -12
