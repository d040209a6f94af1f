cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -5
python train.py -e 1 --resnet-blocks 2 --tile 16 --tiles-per-epoch 16 -b 8 --weights-dir gpurun_out/w 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -1
