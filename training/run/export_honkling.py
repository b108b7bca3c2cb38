from howl_amd.training.run.export_honkling import main

if __name__ == "__main__":
    main()
