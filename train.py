"""CU-Net training on the MI355X hot path with the reference's command line (cu-net.py / options/*.py).

    python train.py --exp_id run1 --exp_dir ./exp --layer_num 2 --order 1 --class_num 16 --loss_num 2 --bs 24 --synthetic 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --exp_id run1 ... --bs 192

See cu_net_amd/driver.py; a real dataset is plugged in by calling cu_net_amd.driver.main(train_loader=..., val_loader=...).
"""
from cu_net_amd.driver import main

if __name__ == '__main__':
    main()
